// vox.hip -- solid voxelization for gfx950 (MI355X).
//
// Result contract: the bitmask of the reference's sequential voxelizer
// (/root/reference/vplib/src/vox/sequential.cpp:6-63), bit for bit.  The reference toggles every
// voxel x in [startX, n) of each covered (y,z) column; here every covered column gets ONE toggle
// at startX and a prefix-XOR along X afterwards fills the rows -- the same map, because XOR is
// associative and commutative.
//
// TILED (reference: vox/tiled.cu:14-576, re-designed as a hybrid):
//   vox_setup    one thread per triangle: per-triangle record with the reference's float
//                expressions (sign, edge deltas, plane A,B,C,D, clamped y/z voxel ranges).
//                SMALL triangles (<= kSmallCells columns in their y/z range -- almost every triangle
//                of a fine mesh) are rasterised right here: one atomicXor per covered column into
//                the toggle grid, no binning at all.  LARGE triangles get their record stored and
//                are counted into the 8x8-column YZ tiles their range overlaps.
//   vox_scan     exclusive scan of the tile histogram (one workgroup).
//   vox_scatter  per-tile lists of large triangles (order inside a tile is irrelevant: XOR commutes).
//   vox_tile     one 256-thread workgroup per tile.  A tile OWNS its 64 x-rows: they are brought
//                into an LDS bit-row buffer (coalesced), the tile's large triangles toggle bits
//                there with ds_xor (no global atomics; records staged through LDS in batches; lane =
//                (y,z) column, wave = triangle slice), and the rows go back to the toggle grid.  Launched for every
//                tile; tiles without large triangles (all of them for a fine mesh) leave at once, so the host never
//                reads a count back and vp_voxelize is fully asynchronous.
//   vox_fill     streaming prefix-XOR of the whole toggle grid into the output.
// NAIVE (reference: vox/naive.cu:12-122): one thread per triangle toggling single bits with
//   global atomicXor, then vox_fill streams the grid once doing the prefix-XOR per row.
//
// All float math below must not be contracted into FMAs: the file is built with
// -ffp-contract=off and carries the pragma as well.
#include "vp_internal.h"

#include <algorithm>
#include <cstdlib>

#pragma clang fp contract(off)

namespace vp {

namespace {

struct F3 { float X, Y, Z; };

__device__ __forceinline__ uint32_t word_prefix_xor(uint32_t v)
{
    v ^= v << 1; v ^= v << 2; v ^= v << 4; v ^= v << 8; v ^= v << 16;
    return v;
}

// Edge tests + plane solve for one (y,z) column centre.  r = record floats 0..15.
// vox/sequential.cpp:47-54 with the orientation sign folded into the edge deltas
// (x*(-1) == -x exactly, and fl(-a - -b) == -fl(a - b), so every comparison is unchanged).
__device__ __forceinline__ bool column_hit(const float* r, float cy, float cz, float ox, float vs,
                                           int n, int& startX)
{
    const float E0 = ((cz - r[1]) * r[2])  - ((cy - r[0]) * r[3]);
    const float E1 = ((cz - r[5]) * r[6])  - ((cy - r[4]) * r[7]);
    const float E2 = ((cz - r[9]) * r[10]) - ((cy - r[8]) * r[11]);
    if (!(E0 >= 0.0f && E1 >= 0.0f && E2 >= 0.0f)) return false;
    const float intersection = (r[15] - (r[13] * cy) - (r[14] * cz)) / r[12];
    const float fx = (intersection - ox) / vs;
    if (!(fx > -2147483648.0f && fx < 2147483648.0f)) return false;   // A == 0: reference is UB
    int sx = (int)fx;
    if (sx < 0) sx = 0;                                                // reference: out-of-bounds
    if (sx >= n) return false;
    startX = sx;
    return true;
}

__device__ __forceinline__ float centre(float o, int i, float vs)
{
    return o + (((float)i * vs) + (vs / 2.0f));                        // vox/sequential.cpp:44-45
}

// Builds the 20-dword record of triangle t; returns false when it covers no column of the slab.
__device__ __forceinline__ bool make_record(const Frame& f, const float* __restrict__ xyz, size_t nverts,
                                            const uint32_t* __restrict__ tri, size_t t, float* r,
                                            int& sy, int& ey, int& sz, int& ez)
{
    const uint32_t i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    if (i0 >= nverts || i1 >= nverts || i2 >= nverts) return false;
    const F3 V0 = { xyz[3 * (size_t)i0], xyz[3 * (size_t)i0 + 1], xyz[3 * (size_t)i0 + 2] };
    const F3 V1 = { xyz[3 * (size_t)i1], xyz[3 * (size_t)i1 + 1], xyz[3 * (size_t)i1 + 2] };
    const F3 V2 = { xyz[3 * (size_t)i2], xyz[3 * (size_t)i2 + 1], xyz[3 * (size_t)i2 + 2] };

    // sequential.cpp:23-24, vox.h:30-32, mesh.h:119-126
    const F3 a = { V1.X - V0.X, V1.Y - V0.Y, V1.Z - V0.Z };
    const F3 b = { V2.X - V1.X, V2.Y - V1.Y, V2.Z - V1.Z };
    const float normalX = (a.Y * b.Z) - (a.Z * b.Y);
    const float s = (normalX >= 0.0f) ? 1.0f : -1.0f;

    // sequential.cpp:26-28, bounding_box.h:31-44
    float minY = V0.Y, maxY = V0.Y, minZ = V0.Z, maxZ = V0.Z;
    if (V1.Y < minY) minY = V1.Y; else if (V1.Y > maxY) maxY = V1.Y;
    if (V1.Z < minZ) minZ = V1.Z; else if (V1.Z > maxZ) maxZ = V1.Z;
    if (V2.Y < minY) minY = V2.Y; else if (V2.Y > maxY) maxY = V2.Y;
    if (V2.Z < minZ) minZ = V2.Z; else if (V2.Z > maxZ) maxZ = V2.Z;

    // sequential.cpp:30-33
    sy = (int)floorf((minY - f.oy) / f.vs);
    ey = (int)ceilf((maxY - f.oy) / f.vs);
    sz = (int)floorf((minZ - f.oz) / f.vs);
    ez = (int)ceilf((maxZ - f.oz) / f.vs);
    sy = max(sy, 0); ey = min(ey, (int)f.n);
    sz = max(sz, (int)f.z0); ez = min(ez, (int)f.z1);

    // sequential.cpp:35-38
    const F3 e1 = { V2.X - V0.X, V2.Y - V0.Y, V2.Z - V0.Z };
    const float A = (a.Y * e1.Z) - (a.Z * e1.Y);
    const float B = (a.Z * e1.X) - (a.X * e1.Z);
    const float C = (a.X * e1.Y) - (a.Y * e1.X);
    const float D = (A * V0.X + B * V0.Y) + C * V0.Z;

    r[0] = V0.Y; r[1] = V0.Z; r[2]  = (V1.Y - V0.Y) * s; r[3]  = (V1.Z - V0.Z) * s;
    r[4] = V1.Y; r[5] = V1.Z; r[6]  = (V2.Y - V1.Y) * s; r[7]  = (V2.Z - V1.Z) * s;
    r[8] = V2.Y; r[9] = V2.Z; r[10] = (V0.Y - V2.Y) * s; r[11] = (V0.Z - V2.Z) * s;
    r[12] = A; r[13] = B; r[14] = C; r[15] = D;
    return sy < ey && sz < ez;
}

// Visits every tile of a triangle's range.  Triangles overlapping many tiles are spread over the
// whole wave so that one huge triangle does not serialise a lane.  Must be reached by all lanes.
template <class Fn>
__device__ __forceinline__ void for_each_tile(bool valid, uint32_t t, int ty0, int ty1, int tz0, int tz1, Fn fn)
{
    const int ny = ty1 - ty0 + 1;
    const int cnt = valid ? ny * (tz1 - tz0 + 1) : 0;
    const bool big = cnt > 16;
    if (!big) {
        for (int i = 0; i < cnt; ++i) fn(t, ty0 + i % ny, tz0 + i / ny);
    }
    unsigned long long m = __ballot(big);
    const int lane = threadIdx.x & 63;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t bt = __shfl(t, src);
        const int by0 = __shfl(ty0, src), bny = __shfl(ny, src), bz0 = __shfl(tz0, src), bcnt = __shfl(cnt, src);
        for (int i = lane; i < bcnt; i += 64) fn(bt, by0 + i % bny, bz0 + i / bny);
    }
}

// ---------------------------------------------------------------------------------------------
#ifndef VP_VOX_SMALL_CELLS
#define VP_VOX_SMALL_CELLS 64       // 16 -> 64: bunny (56 k faces) 0.092 -> 0.053 ms at n = 512, 0.188 -> 0.172 at 1024; finer meshes unchanged (profiles/r02/vox_small.txt)
#endif
constexpr int kSmallCells = VP_VOX_SMALL_CELLS;   // triangles whose y/z voxel range holds at most this many columns skip the binning

__global__ void __launch_bounds__(256)
vox_setup(Frame f, const float* __restrict__ xyz, size_t nverts, const uint32_t* __restrict__ tri, size_t ntris,
          uint4* __restrict__ rec, uint32_t rec_cap, uint32_t* __restrict__ nbig, uint32_t* __restrict__ tile_cnt,
          uint32_t* __restrict__ toggles)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int tilesY = f.n / kTile;
    const int tzBase = f.z0 / kTile;
    float r[16];
    int sy = 0, ey = 0, sz = 0, ez = 0;
    bool valid = false;
    if (t < ntris) valid = make_record(f, xyz, nverts, tri, t, r, sy, ey, sz, ez);
    const int ny = ey - sy, cells = valid ? ny * (ez - sz) : 0;
    const bool small = cells <= kSmallCells;
    // large triangles are rare in fine meshes: their records go to a COMPACT list (small ones write nothing at all --
    // a 16-byte marker per triangle cost 0.4 ms on the 10.8M-face mesh).  The list holds what earlier calls needed (grow-only,
    // the count comes back lazily like the work queue's); a large triangle that finds it full is walked right here, column by
    // column, like a small one: slow for that call, correct for any input, and the next call has the room.
    bool big = valid && !small;
    bool walk = valid && small;
    if (big) {
        const uint32_t slot = atomicAdd(nbig, 1u);                // counts every large triangle, recorded or not
        if (slot >= rec_cap) { big = false; walk = true; }
        else {
            uint4* dst = rec + (size_t)slot * 5;
            dst[0] = make_uint4(__float_as_uint(r[0]),  __float_as_uint(r[1]),  __float_as_uint(r[2]),  __float_as_uint(r[3]));
            dst[1] = make_uint4(__float_as_uint(r[4]),  __float_as_uint(r[5]),  __float_as_uint(r[6]),  __float_as_uint(r[7]));
            dst[2] = make_uint4(__float_as_uint(r[8]),  __float_as_uint(r[9]),  __float_as_uint(r[10]), __float_as_uint(r[11]));
            dst[3] = make_uint4(__float_as_uint(r[12]), __float_as_uint(r[13]), __float_as_uint(r[14]), __float_as_uint(r[15]));
            dst[4] = make_uint4((uint32_t)sy | ((uint32_t)ey << 16), (uint32_t)sz | ((uint32_t)ez << 16), (uint32_t)t, 0u);
        }
    }
    if (walk) {
        for (int i = 0; i < cells; ++i) {                         // same columns, same tests as the tile kernel
            const int y = sy + i % ny, z = sz + i / ny;
            int sx;
            if (column_hit(r, centre(f.oy, y, f.vs), centre(f.oz, z, f.vs), f.ox, f.vs, (int)f.n, sx)) {
                const size_t row = ((size_t)(z - (int)f.z0) * f.n + (size_t)y) * f.w;
                atomicXor(&toggles[row + (sx >> 5)], 1u << (sx & 31));
            }
        }
    }
    const int ty0 = sy / kTile, ty1 = big ? (ey - 1) / kTile : 0;
    const int tz0 = sz / kTile, tz1 = big ? (ez - 1) / kTile : 0;
    for_each_tile(big, (uint32_t)t, ty0, ty1, tz0, tz1, [&](uint32_t, int ty, int tz) {
        atomicAdd(&tile_cnt[(tz - tzBase) * tilesY + ty], 1u);
    });
}

// One workgroup: exclusive scan of cnt[0..m) -> off[0..m], off[m] = total; cur = copy of off.
__global__ void __launch_bounds__(1024)
vox_scan(const uint32_t* __restrict__ cnt, uint32_t m, uint32_t* __restrict__ off, uint32_t* __restrict__ cur)
{
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (m + 1023u) / 1024u;
    const uint32_t b = min(tid * per, m), e = min(b + per, m);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = (tid >= d) ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t i = b; i < e; ++i) { off[i] = run; cur[i] = run; run += cnt[i]; }
    if (tid == 1023) off[m] = part[1023];
}

// One thread per record of the compact large-triangle list.  The list length is only known on the device (*nbig):
// the grid is fixed and strides over it, whole waves at a time (for_each_tile must be reached by all lanes).
__global__ void __launch_bounds__(256)
vox_scatter(Frame f, const uint4* __restrict__ rec, const uint32_t* __restrict__ nbig, uint32_t rec_cap, uint32_t* __restrict__ cur,
            uint32_t* __restrict__ pairs, uint32_t cap)
{
    const size_t nrec = min(*nbig, rec_cap);                       // triangles beyond the list's capacity were walked by vox_setup
    const int tilesY = f.n / kTile;
    const int tzBase = f.z0 / kTile;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); t0 < nrec; t0 += stride) {
        const size_t t = t0 + (threadIdx.x & 63u);
        int sy = 0, ey = 0, sz = 0, ez = 0;
        if (t < nrec) {
            const uint4 q = rec[t * 5 + 4];
            sy = q.x & 0xFFFF; ey = q.x >> 16; sz = q.y & 0xFFFF; ez = q.y >> 16;
        }
        const bool valid = sy < ey && sz < ez;
        const int ty0 = sy / kTile, ty1 = valid ? (ey - 1) / kTile : 0;
        const int tz0 = sz / kTile, tz1 = valid ? (ez - 1) / kTile : 0;
        for_each_tile(valid, (uint32_t)t, ty0, ty1, tz0, tz1, [&](uint32_t tt, int ty, int tz) {
            const uint32_t slot = atomicAdd(&cur[(tz - tzBase) * tilesY + ty], 1u);
            if (slot < cap) pairs[slot] = tt;                      // a tile whose list does not fit scans the record list instead
        });
    }
}

constexpr int kBatch = 64;        // triangle records staged in LDS per round
constexpr int kMaxW = 64;         // words per x-row at n = 2048

// Applies the toggles of the LARGE triangles of one tile to the toggle grid (the small ones were toggled by vox_setup).
// Launched for every tile; a tile without large triangles (every tile of a fine mesh) leaves at once, so the host
// never has to know the list sizes: no read-back, no stream synchronisation inside vp_voxelize.
// `cap` = capacity of `pairs`.  A tile whose slice of the work queue does not fit completely scans the whole record list
// itself (correct for any input; the host grows the queue for the next call from the total it reads back lazily).
__global__ void __launch_bounds__(256)
vox_tile(Frame f, const uint4* __restrict__ rec, const uint32_t* __restrict__ nbig, uint32_t rec_cap, const uint32_t* __restrict__ off,
         const uint32_t* __restrict__ pairs, uint32_t cap, uint32_t* tog)
{
    const int tilesY = f.n / kTile;
    const int tile = blockIdx.x;
    uint32_t begin = off[tile], end = off[tile + 1];
    if (begin == end) return;

    __shared__ uint32_t acc[64 * (kMaxW + 1)];
    __shared__ uint4 srec[kBatch * 5];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = f.w, stride = W + 1;
    const int ty = tile % tilesY, tzl = tile / tilesY;           // tzl: tile row inside the slab
    const bool listed = end <= cap;                               // the tile's slice of the work queue is complete
    if (!listed) { begin = 0; end = min(*nbig, rec_cap); }

    const int rowWords = 64 * W;
    // global word index of LDS row r (= lz*8+ly), word w:  base + (r>>3)*n*W + (r&7)*W + w
    const size_t base = ((size_t)(tzl * kTile) * f.n + (size_t)ty * kTile) * W;
    const size_t planeStride = (size_t)f.n * W;

    // bring the tile's rows of the toggle grid into LDS (row r = lz*8+ly; 8 rows of one lz are contiguous)
    for (int i = tid; i < rowWords; i += 256) {
        const int r = i / W, w = i - r * W;
        acc[r * stride + w] = tog[base + (size_t)(r >> 3) * planeStride + (size_t)(r & 7) * W + w];
    }

    const int y = ty * kTile + (lane & 7);
    const int z = (int)f.z0 + tzl * kTile + (lane >> 3);
    const float cy = centre(f.oy, y, f.vs);
    const float cz = centre(f.oz, z, f.vs);
    uint32_t* myrow = acc + lane * stride;

    for (uint32_t b0 = begin; b0 < end; b0 += kBatch) {
        const int nb = min((uint32_t)kBatch, end - b0);
        __syncthreads();                                          // previous batch consumed
        for (int i = tid; i < nb * 5; i += 256) {
            const int j = i / 5, p = i - j * 5;
            srec[i] = rec[(size_t)(listed ? pairs[b0 + j] : b0 + j) * 5 + p];
        }
        __syncthreads();
        for (int j = wave; j < nb; j += 4) {
            const uint4 q = srec[j * 5 + 4];
            const int sy = q.x & 0xFFFF, ey = q.x >> 16, sz = q.y & 0xFFFF, ez = q.y >> 16;
            if (y >= sy && y < ey && z >= sz && z < ez) {       // also rejects the records of other tiles in a full scan
                float r[16];
                const float4* fr = reinterpret_cast<const float4*>(&srec[j * 5]);
                const float4 r0 = fr[0], r1 = fr[1], r2 = fr[2], r3 = fr[3];
                r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w;
                r[4] = r1.x; r[5] = r1.y; r[6] = r1.z; r[7] = r1.w;
                r[8] = r2.x; r[9] = r2.y; r[10] = r2.z; r[11] = r2.w;
                r[12] = r3.x; r[13] = r3.y; r[14] = r3.z; r[15] = r3.w;
                int sx;
                if (column_hit(r, cy, cz, f.ox, f.vs, (int)f.n, sx))
                    atomicXor(&myrow[sx >> 5], 1u << (sx & 31));
            }
        }
    }
    __syncthreads();

    // coalesced write-back of the toggled rows (the streaming vox_fill does the prefix-XOR for the whole grid afterwards)
    for (int i = tid; i < rowWords; i += 256) {
        const int r = i / W, w = i - r * W;
        tog[base + (size_t)(r >> 3) * planeStride + (size_t)(r & 7) * W + w] = acc[r * stride + w];
    }
}

// ---------------------------------------------------------------------------------------------
// NAIVE: one thread per triangle, single-bit global toggles (vox/naive.cu:12-84 re-stated).
__global__ void __launch_bounds__(256)
vox_naive(Frame f, const float* __restrict__ xyz, size_t nverts, const uint32_t* __restrict__ tri, size_t ntris,
          uint32_t* __restrict__ toggles)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntris) return;
    float r[16];
    int sy, ey, sz, ez;
    if (!make_record(f, xyz, nverts, tri, t, r, sy, ey, sz, ez)) return;
    for (int y = sy; y < ey; ++y) {
        const float cy = centre(f.oy, y, f.vs);
        for (int z = sz; z < ez; ++z) {
            const float cz = centre(f.oz, z, f.vs);
            int sx;
            if (column_hit(r, cy, cz, f.ox, f.vs, (int)f.n, sx)) {
                const size_t row = ((size_t)(z - (int)f.z0) * f.n + (size_t)y) * f.w;
                atomicXor(&toggles[row + (sx >> 5)], 1u << (sx & 31));
            }
        }
    }
}

// Streaming prefix-XOR along x.  One lane = 4 consecutive words (16 B); a row of W words is a
// segment of W/4 lanes; the carry is a segmented XOR-scan of word parities across those lanes.
// dst = (ACC ? dst : 0) ^ fill(src).  src may alias dst when !ACC.
template <bool ACC>
__global__ void __launch_bounds__(256)
vox_fill_vec(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nvec, int lanesPerRow)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // nvec is a multiple of lanesPerRow and blockDim is a multiple of it too, so segments never
    // straddle the grid-stride boundary; out-of-range lanes still take part in the shuffles.
    const size_t rounds = (nvec + stride - 1) / stride;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t rd = 0; rd < rounds; ++rd, i += stride) {
        const bool in = i < nvec;
        uint4 v = in ? src[i] : make_uint4(0, 0, 0, 0);
        const uint32_t px = __popc(v.x) & 1u, py = __popc(v.y) & 1u, pz = __popc(v.z) & 1u, pw = __popc(v.w) & 1u;
        const uint32_t par = px ^ py ^ pz ^ pw;
        uint32_t incl = par;
        for (int d = 1; d < lanesPerRow; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, lanesPerRow);
            if (((int)(threadIdx.x & (lanesPerRow - 1))) >= d) incl ^= o;
        }
        uint32_t c = incl ^ par;                                   // exclusive: parity of the words before
        uint4 o;
        o.x = word_prefix_xor(v.x) ^ (0u - c); c ^= px;
        o.y = word_prefix_xor(v.y) ^ (0u - c); c ^= py;
        o.z = word_prefix_xor(v.z) ^ (0u - c); c ^= pz;
        o.w = word_prefix_xor(v.w) ^ (0u - c);
        if (in) {
            if (ACC) { const uint4 e = dst[i]; o.x ^= e.x; o.y ^= e.y; o.z ^= e.z; o.w ^= e.w; }
            dst[i] = o;
        }
    }
}

// Any other row length (W words, 1 <= W <= 64: n % 128 != 0, or W / 4 not a power of two): one lane = one word, a wave = floor(64 / W) whole
// rows, i.e. 64 - (64 % W) CONSECUTIVE words -- loads and stores are as coalesced as in the vector form.  The carry into a word is the parity
// of the set bits of the words before it in its row: one ballot of the word parities, masked to the lanes [row start, own lane).
// (Round 1 - 4a: one thread per row, W strided words each: 0.029 ms at n = 480 against 0.011 at n = 512.)
template <bool ACC>
__global__ void __launch_bounds__(256)
vox_fill_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t nwords, uint32_t W, size_t nwaves)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t perWave = (64u / W) * W;                        // words a wave handles: whole rows only
    const uint32_t w = lane % W, start = lane - w;                 // word index in its row, first lane of the row
    const size_t wavesPerGrid = (size_t)gridDim.x * (blockDim.x / 64u);
    for (size_t wv = (size_t)blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6); wv < nwaves; wv += wavesPerGrid) {
        const size_t i = wv * perWave + lane;
        const bool in = lane < perWave && i < nwords;
        const uint32_t v = in ? src[i] : 0u;
        const unsigned long long odd = __builtin_amdgcn_ballot_w64((__popc(v) & 1u) != 0u);
        const unsigned long long before = odd & (((1ull << lane) - 1ull) & ~((1ull << start) - 1ull));
        uint32_t o = word_prefix_xor(v) ^ (0u - (uint32_t)(__popcll(before) & 1));
        if (in) {
            if (ACC) o ^= dst[i];
            dst[i] = o;
        }
    }
}

// The toggle grid and the tile histogram start from zero: one launch for both (two hipMemsetAsync calls were two runtime fill kernels,
// ~21 us for the 16 MiB grid of n = 512 and a launch for the 64 KiB histogram: profiles/r04/bench_kernel_stats.csv).
__global__ void __launch_bounds__(256)
vox_zero(uint4* __restrict__ grid, size_t nvec, uint32_t* __restrict__ cnt, uint32_t ncnt)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < nvec; i += stride) grid[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = t0; i < ncnt; i += stride) cnt[i] = 0u;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// dst = (accumulate ? dst : 0) ^ prefix-XOR-along-x(tog); tog may alias dst when !accumulate.
static int launch_fill(vp_ctx* ctx, const Frame& f, const uint32_t* tog, uint32_t* d_words, int accumulate)
{
    hipStream_t st = ctx->stream;
    const size_t nz = f.z1 - f.z0;
    const size_t nwords = (size_t)f.n * f.n * nz / 32;
    ProfScope p(ctx, VP_K_VOX_FILL);
    const int W = f.w;
    const bool vec = (W % 4 == 0) && ((W / 4) & (W / 4 - 1)) == 0 && (W / 4) <= 64;
    if (vec) {
        const size_t nvec = nwords / 4;
        const unsigned blocks = (unsigned)std::min<size_t>((nvec + 255) / 256, 256 * 16);
        if (accumulate)
            hipLaunchKernelGGL(vox_fill_vec<true>, dim3(blocks), dim3(256), 0, st, (const uint4*)tog, (uint4*)d_words, nvec, W / 4);
        else
            hipLaunchKernelGGL(vox_fill_vec<false>, dim3(blocks), dim3(256), 0, st, (const uint4*)tog, (uint4*)d_words, nvec, W / 4);
    } else {
        const size_t perWave = (size_t)(64 / W) * W, nwaves = (nwords + perWave - 1) / perWave;      // W <= 64 (n <= 2048)
        const unsigned blocks = (unsigned)std::min<size_t>((nwaves + 3) / 4, 256 * 16);
        if (accumulate)
            hipLaunchKernelGGL(vox_fill_words<true>, dim3(blocks), dim3(256), 0, st, tog, d_words, nwords, (uint32_t)W, nwaves);
        else
            hipLaunchKernelGGL(vox_fill_words<false>, dim3(blocks), dim3(256), 0, st, tog, d_words, nwords, (uint32_t)W, nwaves);
    }
    return 0;
}

int launch_voxelize(vp_ctx* ctx, const Frame& f, uint32_t* d_words, const float* d_xyz, size_t nverts,
                    const uint32_t* d_tri, size_t ntris, int algo, int accumulate)
{
    hipStream_t st = ctx->stream;
    const size_t nz = f.z1 - f.z0;
    const size_t nwords = (size_t)f.n * f.n * nz / 32;
    const unsigned tblocks = (unsigned)((ntris + 255) / 256);

    // toggle grid: the output itself, or a scratch grid when the result is XORed into existing content
    uint32_t* tog = d_words;
    if (accumulate) {
        VP_TRY(reserve(ctx, ctx->scratch, nwords * 4));
        tog = (uint32_t*)ctx->scratch.ptr;
    }
    // nwords is a multiple of 8 n (n % 32 == 0): whole uint4s; d_words is 16-byte aligned (checked at the ABI), the scratch grid is ours
    auto zero = [&](uint32_t* cnt, uint32_t ncnt) {
        const size_t nvec = nwords / 4;
        const unsigned blocks = (unsigned)std::min<size_t>((nvec + 255) / 256, 256 * 16);
        ProfScope p(ctx, VP_K_VOX_ZERO);
        hipLaunchKernelGGL(vox_zero, dim3(blocks), dim3(256), 0, st, (uint4*)tog, nvec, cnt, ncnt);
    };

    if (algo == VP_ALGO_NAIVE) {
        zero(nullptr, 0);
        if (ntris) {
            ProfScope p(ctx, VP_K_VOX_NAIVE);
            hipLaunchKernelGGL(vox_naive, dim3(tblocks), dim3(256), 0, st, f, d_xyz, nverts, d_tri, ntris, tog);
        }
        VP_TRY(launch_fill(ctx, f, tog, d_words, accumulate));
        VP_HIP(hipGetLastError());
        return 0;
    }

    // ---- TILED (hybrid) ----
    // Every stage is enqueued unconditionally and sized on the host from what earlier calls needed: the record list and the
    // work queue keep that capacity (a large triangle that finds the record list full is walked in place by vox_setup, a tile
    // whose slice of the queue does not fit scans the record list instead), scatter and tile kernels read the list sizes on
    // the device and leave at once when there is nothing to do.  No read-back, no stream synchronisation in steady state.
    const uint32_t tilesY = f.n / kTile;
    const uint32_t numTiles = tilesY * (uint32_t)(nz / kTile);
    VP_TRY(reserve(ctx, ctx->tile_cnt, ((size_t)numTiles + 1) * 4));
    VP_TRY(reserve(ctx, ctx->tile_off, ((size_t)numTiles + 1) * 4));
    VP_TRY(reserve(ctx, ctx->tile_cur, (size_t)numTiles * 4));
    uint32_t* cnt = (uint32_t*)ctx->tile_cnt.ptr;
    uint32_t* off = (uint32_t*)ctx->tile_off.ptr;
    uint32_t* cur = (uint32_t*)ctx->tile_cur.ptr;
    uint32_t* d_nbig = cnt + numTiles;                             // one extra counter after the tile histogram
    if (ntris) {
        // work-queue size and large-triangle count of the previous call, if their copies have landed (never waited for)
        if (ctx->vox_total_event && ctx->vox_total_pending && hipEventQuery(ctx->vox_total_event) == hipSuccess) {
            ctx->vox_total_pending = false;
            ctx->vox_total_seen = std::max<uint64_t>(ctx->vox_total_seen, ctx->vox_total_host[0]);
            ctx->vox_nbig_seen = std::max<uint64_t>(ctx->vox_nbig_seen, ctx->vox_total_host[1]);
            ctx->vox_counts_known = true;
            if (ctx->vox_total_host[1] == 0) ctx->vox_nolist_job = ctx->vox_pending_job;          // that job has no large triangle
            else if (ctx->vox_nolist_job == ctx->vox_pending_job) ctx->vox_nolist_job = vp_ctx::VoxJob{};
        }
        const vp_ctx::VoxJob job{d_tri, ntris, f.n, f.z0, f.z1};
        // A job whose large-triangle count came back as zero (a fine mesh, like the benchmark's) and that is repeated -- the same triangle
        // buffer, count, grid side and slab -- gives the record list no room and leaves the three list kernels out: they would find nothing to do
        // and cost their launches (0.02 ms of a 0.07-ms voxelization at n = 512).  Any other job on the context takes the tile stage (a coarse
        // mesh walked triangle by triangle in vox_setup would cost milliseconds per triangle).  Should the buffer have been refilled in place
        // with large triangles, vox_setup walks them in place (slow for that one call, correct: the path of a full record list), the count
        // comes back non-zero and the job takes the tile stage again.
        const bool noLists = ctx->vox_nolist_job == job && ctx->vox_nolist_job.tri != nullptr;
        const size_t want = std::max<size_t>((size_t)1 << 20, (size_t)ctx->vox_total_seen + ctx->vox_total_seen / 4);
        VP_TRY(reserve(ctx, ctx->pairs, want * 4));
        // record list: 64 Ki records (5 MiB) or what earlier calls needed + 25 %, never more than one per triangle
        const size_t wantRec = std::min<size_t>(ntris, std::max<size_t>((size_t)1 << 16, (size_t)ctx->vox_nbig_seen + ctx->vox_nbig_seen / 4));
        VP_TRY(reserve(ctx, ctx->rec, wantRec * (size_t)kRecDwords * 4));
        uint4* rec = (uint4*)ctx->rec.ptr;
        uint32_t rcap = noLists ? 0u : (uint32_t)std::min<size_t>(ctx->rec.bytes / ((size_t)kRecDwords * 4), ntris);
#ifdef VP_TEST_HOOKS   // test builds only (libvphip_hooks.so): force the walk-in-place path
        if (const char* e = getenv("VP_VOX_REC_CAP")) rcap = std::min<uint32_t>(rcap, (uint32_t)strtoul(e, nullptr, 10));
#endif
        uint32_t* pairs = (uint32_t*)ctx->pairs.ptr;
        uint32_t pcap = (uint32_t)std::min<size_t>(ctx->pairs.bytes / 4, 0xFFFFFFFFu);
#ifdef VP_TEST_HOOKS   // ... and the work-queue overflow path
        if (const char* e = getenv("VP_VOX_QUEUE_CAP")) pcap = std::min<uint32_t>(pcap, (uint32_t)strtoul(e, nullptr, 10));
#endif
        zero(cnt, numTiles + 1u);
        {
            ProfScope p(ctx, VP_K_VOX_SETUP);
            hipLaunchKernelGGL(vox_setup, dim3(tblocks), dim3(256), 0, st, f, d_xyz, nverts, d_tri, ntris, rec, rcap,
                               d_nbig, cnt, tog);
        }
        if (!noLists) {
            ProfScope p(ctx, VP_K_VOX_SCAN);
            hipLaunchKernelGGL(vox_scan, dim3(1), dim3(1024), 0, st, cnt, numTiles, off, cur);
        }
        if (!ctx->vox_total_host) {
            VP_HIP(hipHostMalloc((void**)&ctx->vox_total_host, 2 * sizeof(uint32_t), hipHostMallocDefault));
            ctx->vox_total_host[0] = ctx->vox_total_host[1] = 0;
            VP_HIP(hipEventCreateWithFlags(&ctx->vox_total_event, hipEventDisableTiming));
        }
        if (!ctx->vox_total_pending) {                             // lazily: the next call may grow the queue from it
            if (!noLists) VP_HIP(hipMemcpyAsync(ctx->vox_total_host, off + numTiles, 4, hipMemcpyDeviceToHost, st));   // (no scan: the old figure stays)
            VP_HIP(hipMemcpyAsync(ctx->vox_total_host + 1, d_nbig, 4, hipMemcpyDeviceToHost, st));
            VP_HIP(hipEventRecord(ctx->vox_total_event, st));
            ctx->vox_total_pending = true;
            ctx->vox_pending_job = job;
        }
        if (!noLists) {
            ProfScope p(ctx, VP_K_VOX_SCATTER);
            hipLaunchKernelGGL(vox_scatter, dim3(std::min<unsigned>(tblocks, 1024u)), dim3(256), 0, st, f, rec, d_nbig, rcap, cur, pairs, pcap);
        }
        if (!noLists) {
            ProfScope p(ctx, VP_K_VOX_TILE);
            hipLaunchKernelGGL(vox_tile, dim3(numTiles), dim3(256), 0, st, f, rec, d_nbig, rcap, off, pairs, pcap, tog);
        }
    } else {
        zero(nullptr, 0);                                          // no triangles: an empty grid (or nothing to XOR in)
    }
    VP_TRY(launch_fill(ctx, f, tog, d_words, accumulate));
    VP_HIP(hipGetLastError());
    return 0;
}

}  // namespace vp
