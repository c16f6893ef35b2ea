// jfa_seed.hip -- everything of the JFA that is not the tile kernel: seeding (border mask, init ids), the first pass straight from the
// border mask, the direct kernel of VP_ALGO_NAIVE, the small-grid table kernel, the id -> sdf conversion, and the launchers / dispatch.
//
// Kernels (templated on the id format, jfa_common.h)
//   jfa_border_march  bitmask -> border bitmask: a lane owns a word column and marches along z.
//   jfa_init          bitmask -> ids (and / or the border bitmask).  One lane = one 32-voxel word: the 26-neighbourhood test is 27 word
//                     loads + shifts / ANDs; ids leave as coalesced 16-byte stores after a wave shuffle transposes word-per-lane into
//                     voxels-per-lane.
//   jfa_first_pass    step k = n/2 straight from the border bitmask (no init id volume); n % 128 == 0.
//   jfa_pass_direct   (VP_ALGO_NAIVE) one thread per voxel, everything recomputed inline: the independent form the tile kernels are
//                     tested against, pass by pass.
//   jfa_pass_table    n < 96: LDS coordinate tables, one workgroup per few rows.
//   jfa_final         ids + bitmask -> float sdf.
#include "jfa_common.h"

namespace vp {
namespace {

// ------------------------------------------------------------------------------------------ init
// words: slab bitmask; below/above: plane z0-1 / z1 (or null).  Returns the word holding voxels
// (32*xw .., y, zg) or 0 outside the grid (outside counts as unset, sequential.cpp:46-51).
__device__ __forceinline__ uint32_t grid_word(const Frame& f, const uint32_t* __restrict__ words,
                                              const uint32_t* __restrict__ below, const uint32_t* __restrict__ above,
                                              int xw, int y, int zg)
{
    if (xw < 0 || xw >= (int)f.w || y < 0 || y >= (int)f.n || zg < 0 || zg >= (int)f.n) return 0u;
    const size_t inPlane = (size_t)y * f.w + xw;
    if (zg < (int)f.z0) return (below != nullptr && zg == (int)f.z0 - 1) ? below[inPlane] : 0u;
    if (zg >= (int)f.z1) return (above != nullptr && zg == (int)f.z1) ? above[inPlane] : 0u;
    return words[(size_t)(zg - (int)f.z0) * f.n * f.w + inPlane];
}

__device__ __forceinline__ void store4(uint32_t* base, size_t quad, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    reinterpret_cast<uint4*>(base)[quad] = make_uint4(a, b, c, d);
}
__device__ __forceinline__ void store4(uint2* base, size_t quad, uint2 a, uint2 b, uint2 c, uint2 d)
{
    uint4* p = reinterpret_cast<uint4*>(base) + quad * 2;
    p[0] = make_uint4(a.x, a.y, b.x, b.y);
    p[1] = make_uint4(c.x, c.y, d.x, d.y);
}
// CPT (ID = Id64 inside the kernel): the ids leave in the compact layout of a window -- `ids` = word plane z0, `idsB` = byte plane z0.
template <class ID, bool IDS, bool MASK, bool CPT = false>
__global__ void __launch_bounds__(256)
jfa_init(Frame f, const uint32_t* __restrict__ words, const uint32_t* __restrict__ below,
         const uint32_t* __restrict__ above, typename ID::T* __restrict__ ids, unsigned char* __restrict__ idsB, uint32_t* __restrict__ border_words)
{
    using T = typename ID::T;
    const int lane = threadIdx.x & 63;
    const size_t wbase = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;   // first word of this wave
    const size_t wi = wbase + lane;
    const int W = f.w;
    const int xw = (int)(wi % W);
    const size_t row = wi / W;
    const int y = (int)(row % f.n);
    const int zg = (int)(row / f.n) + (int)f.z0;

    const uint32_t centre = grid_word(f, words, below, above, xw, y, zg);
    uint32_t border = 0;
    // Rows of a power-of-two number of words (n = 32, 64, ..., 2048) never straddle a wave, so the words left and right of
    // a lane's word sit in the neighbouring lanes: 9 loads + 18 lane shuffles per word instead of 27 loads (wave-uniform
    // skip of empty waves keeps every lane in the shuffles).
    const bool pow2 = (W & (W - 1)) == 0;
    if (pow2 ? __any(centre != 0u) : (centre != 0u)) {
        uint32_t interior = 0xFFFFFFFFu;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                const uint32_t c = (dz == 0 && dy == 0) ? centre : grid_word(f, words, below, above, xw, y + dy, zg + dz);
                uint32_t p, n;
                if (pow2) {
                    p = (uint32_t)__shfl_up((int)c, 1);
                    n = (uint32_t)__shfl_down((int)c, 1);
                    if (xw == 0) p = 0u;                            // outside the grid counts as unset (sequential.cpp:46-51)
                    if (xw == W - 1) n = 0u;
                } else {
                    p = grid_word(f, words, below, above, xw - 1, y + dy, zg + dz);
                    n = grid_word(f, words, below, above, xw + 1, y + dy, zg + dz);
                }
                const uint32_t left = (c << 1) | (p >> 31);       // bit i = voxel x-1
                const uint32_t right = (c >> 1) | (n << 31);      // bit i = voxel x+1
                interior &= left & c & right;
            }
        border = centre & ~interior;                              // sequential.cpp:28-55
    }
    if (MASK) border_words[wi] = border;
    if (IDS) {
        const T mybase = ID::pack((uint32_t)xw * 32u, (uint32_t)y, (uint32_t)zg);
        const int sub = (lane & 7) * 4;
        T* out = ids + wbase * 32;
#pragma unroll 4
        for (int j = 0; j < 8; ++j) {
            const int src = j * 8 + (lane >> 3);
            const uint32_t b = (__shfl(border, src) >> sub) & 0xFu;
            const T id0 = ID::add_x(ID::shfl(mybase, src), (uint32_t)sub);
            const T v0 = ID::sel(b & 1u, id0, ID::none()), v1 = ID::sel(b & 2u, ID::add_x(id0, 1u), ID::none());
            const T v2 = ID::sel(b & 4u, ID::add_x(id0, 2u), ID::none()), v3 = ID::sel(b & 8u, ID::add_x(id0, 3u), ID::none());
            if constexpr (CPT) {
                const uint2 c0 = IdC::from64(v0), c1 = IdC::from64(v1), c2 = IdC::from64(v2), c3 = IdC::from64(v3);
                const size_t quad = wbase * 8 + (size_t)j * 64 + lane;             // four voxels: 16 bytes of words, 4 bytes of the byte plane
                reinterpret_cast<uint4*>(ids)[quad] = make_uint4(c0.x, c1.x, c2.x, c3.x);
                reinterpret_cast<uint32_t*>(idsB)[quad] = c0.y | (c1.y << 8) | (c2.y << 16) | (c3.y << 24);
            } else {
                store4(out, (size_t)j * 64 + lane, v0, v1, v2, v3);
            }
        }
    }
}

// Border mask alone (vp_surface, the "::Initialization" half of vp_jfa), rows of up to 64 words (every legal n).
// A lane owns one word column (xw, y) and MARCHES along z over `zc` planes.  Per plane it forms
//     H(z) = AND over the rows y-1, y, y+1 of (left & word & right)            -- the 3 x 3 in-plane part of the 26-neighbourhood
// from three word loads (the left / right words come from the neighbouring lanes: v_mov_b32_dpp wave_shr / wave_shl, a VALU
// move instead of the ds_bpermute of __shfl), keeps the H of three consecutive planes in registers, and
//     border(z) = word(z) & ~(H(z-1) & H(z) & H(z+1))                            (sequential.cpp:28-55).
// 3 (zc + 2) / zc word loads and 6 lane moves per output word where jfa_init needs 9 and 18, and zc times fewer, longer
// workgroups (jfa_init at n = 1024: 131,072 workgroups of 256 words, 0.23 ms for 2 x 128 MiB = 1.1 TB/s).
__global__ void __launch_bounds__(256)
jfa_border_march(Frame f, const uint32_t* __restrict__ words, const uint32_t* __restrict__ below,
                 const uint32_t* __restrict__ above, uint32_t* __restrict__ border_words, uint32_t zc)
{
    const int W = (int)f.w, N = (int)f.n;
    // A wave holds floor(64 / W) WHOLE rows (all 64 lanes when W divides 64: every power-of-two side), so the left / right word of a lane
    // is always in the neighbouring lane; the lanes past the last whole row idle.  (Until late in round 4 only power-of-two W >= 4 ran here.)
    const uint32_t lane = threadIdx.x & 63u, rowsPerWave = 64u / (uint32_t)W;
    const uint32_t wv = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int xw = (int)(lane % (uint32_t)W);
    const int y = (int)(wv * rowsPerWave + lane / (uint32_t)W);
    const bool valid = lane < rowsPerWave * (uint32_t)W && y < N;
    const uint32_t wi = (uint32_t)y * (uint32_t)W + (uint32_t)xw;  // word index inside a plane
    const int zfirst = (int)f.z0 + (int)(blockIdx.y * zc);
    const int zlast = min(zfirst + (int)zc, (int)f.z1);           // exclusive
    const size_t planeWords = (size_t)N * W;
    const bool xlo = xw == 0, xhi = xw == W - 1;
    // in-plane part of plane zg; `centre` receives the lane's own word
    auto inplane = [&](int zg, uint32_t& centre) -> uint32_t {
        uint32_t r[3];
        if (zg < 0 || zg >= N) { centre = 0u; return 0u; }         // outside the grid counts as unset (sequential.cpp:46-51); wave-uniform
        const uint32_t* pl = zg < (int)f.z0 ? (zg == (int)f.z0 - 1 ? below : nullptr)
                           : zg >= (int)f.z1 ? (zg == (int)f.z1 ? above : nullptr)
                           : words + (size_t)(zg - (int)f.z0) * planeWords;
        if (pl == nullptr) { centre = 0u; return 0u; }              // a halo plane the caller did not give: as outside
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            r[dy + 1] = (valid && yy >= 0 && yy < N) ? pl[(size_t)yy * W + xw] : 0u;
        }
        centre = r[1];
        uint32_t h = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            uint32_t p = lane_prev(r[j]), n = lane_next(r[j]);
            if (xlo) p = 0u;
            if (xhi) n = 0u;
            h &= ((r[j] << 1) | (p >> 31)) & r[j] & ((r[j] >> 1) | (n << 31));
        }
        return h;
    };
    uint32_t cPrev, cCur, cNext;
    uint32_t hPrev = inplane(zfirst - 1, cPrev);
    uint32_t hCur = inplane(zfirst, cCur);
    for (int zg = zfirst; zg < zlast; ++zg) {
        const uint32_t hNext = inplane(zg + 1, cNext);
        if (valid) border_words[(size_t)(zg - (int)f.z0) * planeWords + wi] = cCur & ~(hPrev & hCur & hNext);
        hPrev = hCur; hCur = hNext; cCur = cNext;
    }
}
// ------------------------------------------------------------------------------------------ pass
// Plane of global z `zg` among the three id buffers of a slab (see vphip.h, vp_jfa_pass).
template <class T>
__device__ __forceinline__ const T* id_plane(const Frame& f, uint32_t k, const T* in, const T* minus, const T* plus, int zg)
{
    const size_t plane = (size_t)f.n * f.n;
    if (zg < (int)f.z0) return minus + (size_t)(zg - ((int)f.z0 - (int)k)) * plane;
    if (zg >= (int)f.z1) {
        const int pbase = max((int)f.z1, (int)f.z0 + (int)k);
        return plus + (size_t)(zg - pbase) * plane;
    }
    return in + (size_t)(zg - (int)f.z0) * plane;
}

template <class ID>
__global__ void __launch_bounds__(256)
jfa_pass_direct(Frame f, uint32_t k, const typename ID::T* __restrict__ in, const typename ID::T* __restrict__ minus,
                const typename ID::T* __restrict__ plus, typename ID::T* __restrict__ out)
{
    using T = typename ID::T;
    // grid = (n*n/256, planes): n*n is a multiple of 1024, and 2-D keeps the thread count per dimension < 2^32 at n = 2048
    const uint32_t inPlane = blockIdx.x * 256u + threadIdx.x;
    const size_t gid = (size_t)blockIdx.y * f.n * f.n + inPlane;
    const int N = (int)f.n;
    const int x = (int)(inPlane % f.n);
    const int y = (int)(inPlane / f.n);
    const int zg = (int)blockIdx.y + (int)f.z0;
    const float px = axis_pos(f.ox, x, f.vs), py = axis_pos(f.oy, y, f.vs), pz = axis_pos(f.oz, zg, f.vs);

    T best = in[gid];
    float bestd = ID::is_none(best) ? INFINITY : seed_distance<ID>(f, best, px, py, pz);   // = fabs(sdf), :84
    for (int dz = -1; dz <= 1; ++dz) {
        const int nz = zg + dz * (int)k;
        if (nz < 0 || nz >= N) continue;
        const T* pl = id_plane(f, k, in, minus, plus, nz);
        for (int dy = -1; dy <= 1; ++dy) {
            const int ny = y + dy * (int)k;
            if (ny < 0 || ny >= N) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                if (dx == 0 && dy == 0 && dz == 0) continue;
                const int nx = x + dx * (int)k;
                if (nx < 0 || nx >= N) continue;
                const T c = pl[(size_t)ny * N + nx];
                if (!ID::is_none(c)) {                             // fabs(seed) < INFINITY, :102
                    const float d = seed_distance<ID>(f, c, px, py, pz);
                    if (d < bestd) { bestd = d; best = c; }        // :106-110
                }
            }
        }
    }
    out[gid] = best;
}
// First pass (k = n/2) straight from the border bitmask.  Before any pass the state is trivial: a border
// voxel's seed is itself, everything else is none (sequential.cpp:55-60), so the first pass needs no id
// volume at all -- a candidate exists iff its border bit is set and its id is its own coordinates.  This
// drops the id volume jfa_init would write and this pass would read back.
// One wave = one 64-voxel x-segment.  Requires n % 128 == 0, so k is a multiple of 64 and every candidate
// segment of a wave is exactly two aligned mask words.  With k = n/2 exactly one of -k / +k is inside the grid per axis,
// wave-uniformly: 8 candidate segments (the own one + 7), not 27.  Lane q < 8 fetches the words of segment q -- ONE vector
// load instruction per wave and row (one scalar load per segment was measured 10x slower: the scalar cache thrashes) --
// and v_readlane distributes the masks as wave-uniform values, so empty segments are skipped with scalar branches.  The
// per-axis squared differences and id parts are formed once per wave / row, a candidate costs two adds and the
// compare + selects.  (Round 1 walked all 27 candidate slots with per-candidate index arithmetic on the scalar unit:
// 43 SALU + 57 VALU per row and the CU's scalar unit 66 % busy; profiles/r01.)
// `border` is the border mask of the WHOLE grid (vp_surface); the kernel produces the planes of `f`.
constexpr int kFirstRows = 16;                // rows per wave: their mask loads are all in flight before the first is used

// CPT as in jfa_init.
template <class ID, bool CPT = false>
__global__ void __launch_bounds__(256)
jfa_first_pass(Frame f, uint32_t k, const uint32_t* __restrict__ border, typename ID::T* __restrict__ out, unsigned char* __restrict__ outB, uint32_t gx, uint32_t gy)
{
    // one-dimensional launch (see jfa_first_two): x block fastest, then row block, then plane
    const uint32_t bIdxX = blockIdx.x % gx, bIdxY = (blockIdx.x / gx) % gy, bIdxZ = blockIdx.x / (gx * gy);
    using T = typename ID::T;
    const int N = (int)f.n;
    const int K = (int)k;                                           // = n / 2: per axis exactly one of -k / +k is inside the grid
    const int lane = threadIdx.x & 63;
    const int x0 = __builtin_amdgcn_readfirstlane((int)(bIdxX * 256u + (threadIdx.x & ~63u)));   // segment start
    if (x0 >= N) return;                                            // whole wave
    const int x = x0 + lane;
    const int ybase = bIdxY * kFirstRows;
    const int zl = bIdxZ;
    const int zg = zl + (int)f.z0;
    // The in-grid neighbour along each axis (wave-uniform: k is a multiple of 64, the rows of a wave are 8-aligned): 8
    // candidate segments in all -- the own one and 7 others -- instead of the 27 of a general pass.
    const int ax = x0 < K ? K : -K, ay = ybase < K ? K : -K, az = zg < K ? K : -K;

    // Lane L < 8 fetches the two mask words of the L-th candidate segment IN SCAN ORDER (z, y, x; sequential.cpp:86-88):
    // bit 2 / 1 / 0 of L = second position along z / y / x, where the first position is the neighbour if it lies at -k and
    // the voxel's own coordinate otherwise.  s = 1 marks the neighbour.
    const int sx0 = ax < 0, sy0 = ay < 0, sz0 = az < 0;               // is the FIRST position along the axis the neighbour?
    const int bxL = (lane & 1) ^ sx0, byL = ((lane >> 1) & 1) ^ sy0, bzL = ((lane >> 2) & 1) ^ sz0;
    const int ownLane = sz0 * 4 + sy0 * 2 + sx0;                       // the voxel's own segment (all three on "own")
    uint2 mine[kFirstRows];
#pragma unroll
    for (int r = 0; r < kFirstRows; ++r) {
        mine[r] = make_uint2(0u, 0u);
        if (lane < 8)
            mine[r] = *reinterpret_cast<const uint2*>(border + ((((size_t)(zg + bzL * az) * N + (ybase + r + byL * ay)) * N + (x0 + bxL * ax)) >> 5));
    }
    // Per axis and position: squared coordinate difference (0 to oneself: fl(p - p) = 0 exactly) and id part -- the
    // reference's expressions (jfa/jfa.h:19-20, sequential.cpp:79-81), evaluated once per wave / row instead of per candidate.
    const float px = axis_pos(f.ox, x, f.vs), pz = axis_pos(f.oz, zg, f.vs);
    const float ddxv = axis_pos(f.ox, x + ax, f.vs) - px, ddzv = axis_pos(f.oz, zg + az, f.vs) - pz;
    const float dxx = ddxv * ddxv, dzz = ddzv * ddzv;
    const float dxs[2] = {sx0 ? dxx : 0.0f, sx0 ? 0.0f : dxx}, dzs[2] = {sz0 ? dzz : 0.0f, sz0 ? 0.0f : dzz};
    const T idxOwn = ID::pack((uint32_t)x, 0u, 0u), idxNb = ID::pack((uint32_t)(x + ax), 0u, 0u);
    const T idzOwn = ID::pack(0u, 0u, (uint32_t)zg), idzNb = ID::pack(0u, 0u, (uint32_t)(zg + az));
    const T idxs[2] = {ID::sel(sx0, idxNb, idxOwn), ID::sel(sx0, idxOwn, idxNb)};
    const T idzs[2] = {ID::sel(sz0, idzNb, idzOwn), ID::sel(sz0, idzOwn, idzNb)};
#pragma unroll
    for (int r = 0; r < kFirstRows; ++r) {
        const int y = ybase + r;
        const unsigned long long own = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine[r].x, ownLane) |
                                       ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine[r].y, ownLane) << 32);
        const float py = axis_pos(f.oy, y, f.vs);
        const float ddyv = axis_pos(f.oy, y + ay, f.vs) - py;
        const float dyy = ddyv * ddyv;
        const float dys[2] = {sy0 ? dyy : 0.0f, sy0 ? 0.0f : dyy};
        const T idyOwn = ID::pack(0u, (uint32_t)y, 0u), idyNb = ID::pack(0u, (uint32_t)(y + ay), 0u);
        const T idys[2] = {ID::sel(sy0, idyNb, idyOwn), ID::sel(sy0, idyOwn, idyNb)};
        T best = ID::none();
        float bestd = INFINITY;
        if ((own >> lane) & 1ull) { best = ID::join(ID::join(idxOwn, idyOwn), idzOwn); bestd = 0.0f; }   // own seed: distance 0 (:56)
        // one ballot = the set of neighbour segments that hold a border voxel at all; only those are evaluated, in lane order
        const uint32_t todo = (uint32_t)__ballot(lane < 8 && lane != ownLane && (mine[r].x | mine[r].y) != 0u);
        if (todo) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (!((todo >> c) & 1u)) continue;                  // wave-uniform
                const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)mine[r].x, c), mhi = (uint32_t)__builtin_amdgcn_readlane((int)mine[r].y, c);
                const unsigned long long m = (unsigned long long)mlo | ((unsigned long long)mhi << 32);
                const bool has = (m >> lane) & 1ull;
                const float d = (dxs[c & 1] + dys[(c >> 1) & 1]) + dzs[(c >> 2) & 1];                     // jfa/jfa.h:19-20
                const T id = ID::join(ID::join(idxs[c & 1], idys[(c >> 1) & 1]), idzs[(c >> 2) & 1]);
                const bool take = has & (d < bestd);
                bestd = take ? d : bestd;
                best = ID::sel(take, id, best);
            }
        }
        const size_t vox = ((size_t)zl * N + y) * N + x;
        if constexpr (CPT) {
            const uint2 c = IdC::from64(best);
            reinterpret_cast<uint32_t*>(out)[vox] = c.x;
            outB[vox] = (unsigned char)c.y;
        } else {
            out[vox] = best;
        }
    }
}
constexpr int kTableKernelTab = 1024;   // entries per table of jfa_pass_table (every field offset of Id9, "none" included, stays inside)

// Table variant for small grids (n < kTileMinN = 96, 32-bit ids).  Workgroup = RY consecutive x-rows of one z.
// LDS: PX[i] = ox + i*vs; TZ[i] = (PZ[i]-pz)^2 for this z; TY[r][i] = (PY[i]-py_r)^2 for row r.
// dist = ((PX[ix]-px)^2 + TY[iy]) + TZ[iz]  -- the same float operations as seed_distance().
__global__ void __launch_bounds__(256)
jfa_pass_table(Frame f, uint32_t k, const uint32_t* __restrict__ in, const uint32_t* __restrict__ minus,
               const uint32_t* __restrict__ plus, uint32_t* __restrict__ out, int RY)
{
    using ID = Id9;                                                // n < kTileMinN
    constexpr int kTab = kTableKernelTab;
    extern __shared__ float lds[];
    float* PX = lds;
    float* TZ = lds + kTab;
    float* TY = lds + 2 * kTab;

    const int N = (int)f.n;
    const int tid = threadIdx.x;
    const int zl = (int)blockIdx.y;
    const int zg = zl + (int)f.z0;
    const int y0 = blockIdx.x * RY;
    const float pz = axis_pos(f.oz, zg, f.vs);

    for (int i = tid; i < N; i += 256) {
        PX[i] = axis_pos(f.ox, i, f.vs);
        const float dzv = axis_pos(f.oz, i, f.vs) - pz;
        TZ[scr(i)] = dzv * dzv;
        const float sy = axis_pos(f.oy, i, f.vs);
        for (int r = 0; r < RY; ++r) {
            const float dyv = sy - axis_pos(f.oy, y0 + r, f.vs);
            TY[r * kTab + scr(i)] = dyv * dyv;
        }
    }
    __syncthreads();

    const int r = tid / N, xs = tid - r * N;
    if (r >= RY) return;
    const int y = y0 + r;
    if (y >= N) return;
    const char* ty = reinterpret_cast<const char*>(TY + r * kTab);
    const char* tz = reinterpret_cast<const char*>(TZ);
    const char* tx = reinterpret_cast<const char*>(PX);

    // the (up to) 9 source rows; null = outside the grid
    const uint32_t* rows[9];
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz) {
        const int nz = zg + dz * (int)k;
        const bool zin = nz >= 0 && nz < N;
        const uint32_t* pl = zin ? id_plane(f, k, in, minus, plus, nz) : nullptr;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int ny = y + dy * (int)k;
            rows[(dz + 1) * 3 + (dy + 1)] = (zin && ny >= 0 && ny < N) ? pl + (size_t)ny * N : nullptr;
        }
    }
    uint32_t* orow = out + ((size_t)zl * N + y) * N;

    for (int x = xs; x < N; x += N) {
        const float px = PX[x];
        const int xm = x - (int)k, xp = x + (int)k;
        const bool hasM = xm >= 0, hasP = xp < N;

        uint32_t c[27];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const uint32_t* rw = rows[q];
            c[q * 3 + 0] = (rw && hasM) ? rw[xm] : ID::none();
            c[q * 3 + 1] = rw ? rw[x] : ID::none();
            c[q * 3 + 2] = (rw && hasP) ? rw[xp] : ID::none();
        }
        uint32_t best = c[13];
        float bestd = INFINITY;
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            // own state first (it wins ties: acceptance is strict, sequential.cpp:106), then scan order
            const int q = (j == 0) ? 13 : (j <= 13 ? j - 1 : j);
            const uint32_t id = c[q];
            const float sx = *reinterpret_cast<const float*>(tx + ID::xoff(id));
            const float dy2 = *reinterpret_cast<const float*>(ty + ID::yoff(id));
            const float dz2 = *reinterpret_cast<const float*>(tz + ID::zoff(id));
            const float dxv = sx - px;
            const float d = ((dxv * dxv) + dy2) + dz2;
            const bool take = !ID::is_none(id) && (d < bestd);
            bestd = take ? d : bestd;
            best = take ? id : best;
        }
        orow[x] = best;
    }
}
// ------------------------------------------------------------------------------------------ final
// One lane = 4 voxels.  sequential.cpp:55-60,106-109 + apps/cli/main.cpp:200 give the sign rule.
__device__ __forceinline__ void load4(const uint32_t* base, size_t quad, uint32_t (&o)[4])
{
    const uint4 v = reinterpret_cast<const uint4*>(base)[quad];
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void load4(const uint2* base, size_t quad, uint2 (&o)[4])
{
    const uint4* p = reinterpret_cast<const uint4*>(base) + quad * 2;
    const uint4 a = p[0], b = p[1];
    o[0] = make_uint2(a.x, a.y); o[1] = make_uint2(a.z, a.w); o[2] = make_uint2(b.x, b.y); o[3] = make_uint2(b.z, b.w);
}

template <class ID>
__global__ void __launch_bounds__(256)
jfa_final(Frame f, const uint32_t* __restrict__ words, const typename ID::T* __restrict__ ids, float fill,
          float4* __restrict__ sdf)
{
    using T = typename ID::T;
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total4 = (size_t)f.n * f.n * (f.z1 - f.z0) / 4;
    if (i4 >= total4) return;
    const size_t v = i4 * 4;
    const uint32_t x = (uint32_t)(v % f.n);
    const uint32_t y = (uint32_t)((v / f.n) % f.n);
    const uint32_t zg = (uint32_t)(v / ((size_t)f.n * f.n)) + f.z0;
    const uint32_t bits = (words[v >> 5] >> (v & 31)) & 0xFu;
    T idv[4];
    load4(ids, i4, idv);
    const float py = axis_pos(f.oy, y, f.vs), pz = axis_pos(f.oz, zg, f.vs);
    float o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const bool set = (bits >> b) & 1u;
        const float init = set ? INFINITY : fill;                  // interior +inf (:59) / caller's fill
        if (ID::is_none(idv[b])) { o[b] = init; continue; }
        const float d = seed_distance<ID>(f, idv[b], axis_pos(f.ox, x + b, f.vs), py, pz);
        o[b] = copysignf(d, init);                                 // :108
    }
    sdf[i4] = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

size_t jfa_id_bytes(const Frame& f) { return wide(f) ? 8 : 4; }

// ------------------------------------------------------------------------------------------ rows of "none"
// One row of "none" per id format for out-of-grid reads of the tile kernel: 1024 x 4 bytes of each 32-bit "none", then the compact
// format's word row (2048 x 4) with its byte row (2048 x 1) right behind it.
int ensure_none_rows(vp_ctx* ctx)
{
    if (ctx->none_row.ptr) return 0;
    VP_TRY(reserve(ctx, ctx->none_row, 2 * 1024 * 4 + 2048 * 4 + 2048));
    char* p = (char*)ctx->none_row.ptr;
    VP_HIP(hipMemsetD32Async((hipDeviceptr_t)p, (int)kNone9, 1024, ctx->stream));
    VP_HIP(hipMemsetD32Async((hipDeviceptr_t)(p + 1024 * 4), (int)kNone10, 1024, ctx->stream));
    VP_HIP(hipMemsetD32Async((hipDeviceptr_t)(p + 2 * 1024 * 4), (int)IdC::kNoneWord, 2048, ctx->stream));
    VP_HIP(hipMemsetAsync(p + 2 * 1024 * 4 + 2048 * 4, (int)IdC::kNoneByte, 2048, ctx->stream));
    return 0;
}
const void* none_row_id9(vp_ctx* ctx) { return ctx->none_row.ptr; }
const void* none_row_id10(vp_ctx* ctx) { return (const char*)ctx->none_row.ptr + 1024 * 4; }
const void* none_row_idc(vp_ctx* ctx) { return (const char*)ctx->none_row.ptr + 2 * 1024 * 4; }
static_assert(Id9::kNoneValue == kNone9 && Id10::kNoneValue == kNone10, "vp_internal.h");

// ------------------------------------------------------------------------------------------ seeding
// d_ids != nullptr: init ids (plain: 4 / 8 bytes per voxel) for the planes of f, optionally the border mask too; d_ids == nullptr: the
// border mask alone (vp_surface, the "::Initialization" half of vp_jfa).
int launch_jfa_init(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* below,
                    const uint32_t* above, void* d_ids, uint32_t* d_border_words)
{
    const size_t nwords = (size_t)f.n * f.n * (f.z1 - f.z0) / 32;
    const unsigned blocks = (unsigned)(nwords / 256);             // nwords is a multiple of 256
    ProfScope p(ctx, d_ids ? VP_K_JFA_INIT : VP_K_SURFACE);
    if (!d_ids && d_border_words) {
        // border mask alone: lanes march along z (jfa_border_march; rows of up to 64 words: every legal n); chunks of zc planes, short
        // enough to fill the chip
        const uint32_t rowsPerWave = 64u / f.w, wavesPerPlane = (f.n + rowsPerWave - 1) / rowsPerWave;
        const uint32_t inPlane = (wavesPerPlane + 3u) / 4u, nz = f.z1 - f.z0;
        uint32_t zc = 32;
        while (zc > 4 && inPlane * ((nz + zc - 1) / zc) < 8u * (uint32_t)ctx->cus) zc /= 2;
        hipLaunchKernelGGL(jfa_border_march, dim3(inPlane, (nz + zc - 1) / zc), dim3(256), 0, ctx->stream, f, d_words, below, above, d_border_words, zc);
        VP_HIP(hipGetLastError());
        return 0;
    }
#define VP_INIT(ID, M) hipLaunchKernelGGL((jfa_init<ID, true, M>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, \
                                          (typename ID::T*)d_ids, (unsigned char*)nullptr, d_border_words)
    if (wide(f))         { if (d_border_words) VP_INIT(Id64, true); else VP_INIT(Id64, false); }
    else if (f.n <= 512) { if (d_border_words) VP_INIT(Id9, true);  else VP_INIT(Id9, false); }
    else                 { if (d_border_words) VP_INIT(Id10, true); else VP_INIT(Id10, false); }
#undef VP_INIT
    VP_HIP(hipGetLastError());
    return 0;
}

// init ids of the planes of f into a window (the library's layout: compact above n = 1024)
int launch_win_init(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* below, const uint32_t* above, const IdWin& out)
{
    const size_t nwords = (size_t)f.n * f.n * (f.z1 - f.z0) / 32;
    const unsigned blocks = (unsigned)(nwords / 256);
    ProfScope p(ctx, VP_K_JFA_INIT);
    char* w = win_words(out, f.n, out.at);
    if (win_compact(f.n))
        hipLaunchKernelGGL((jfa_init<Id64, true, false, true>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, (uint2*)w,
                           (unsigned char*)win_bytes_plane(out, f.n, out.at), (uint32_t*)nullptr);
    else if (f.n <= 512)
        hipLaunchKernelGGL((jfa_init<Id9, true, false>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, (uint32_t*)w, (unsigned char*)nullptr, (uint32_t*)nullptr);
    else
        hipLaunchKernelGGL((jfa_init<Id10, true, false>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, (uint32_t*)w, (unsigned char*)nullptr, (uint32_t*)nullptr);
    VP_HIP(hipGetLastError());
    return 0;
}

bool jfa_can_start_from_mask(const Frame& f, int algo) { return algo == VP_ALGO_TILED && f.n >= kTileMinN && f.n % 128 == 0; }

// The fused start (passes n/2 and n/4 in one launch from the border mask, jfa_first_two.hip) serves every whole grid the tile kernels serve.
bool jfa_can_fuse_first_two(const Frame& f, int algo) { return algo == VP_ALGO_TILED && f.n >= kTileMinN && f.z0 == 0 && f.z1 == f.n; }

// First pass (k = n/2) of the planes of f from the whole-grid border mask into a window (see jfa_first_pass).
int launch_win_first_pass(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, const IdWin& out)
{
    ProfScope p(ctx, VP_K_JFA_FIRST);
    const uint32_t gx = (f.n + 255) / 256, gy = f.n / kFirstRows;
    const dim3 grid(gx * gy * (f.z1 - f.z0));
    char* w = win_words(out, f.n, out.at);
    if (win_compact(f.n)) hipLaunchKernelGGL((jfa_first_pass<Id64, true>), grid, dim3(256), 0, ctx->stream, f, f.n / 2, d_border, (uint2*)w, (unsigned char*)win_bytes_plane(out, f.n, out.at), gx, gy);
    else if (f.n <= 512)  hipLaunchKernelGGL((jfa_first_pass<Id9>), grid, dim3(256), 0, ctx->stream, f, f.n / 2, d_border, (uint32_t*)w, (unsigned char*)nullptr, gx, gy);
    else                  hipLaunchKernelGGL((jfa_first_pass<Id10>), grid, dim3(256), 0, ctx->stream, f, f.n / 2, d_border, (uint32_t*)w, (unsigned char*)nullptr, gx, gy);
    VP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ passes
// One pass on a window (tile kernel).  Timing key per variant: their algorithmic bytes differ.
int launch_win_pass(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t stride,
                    const uint32_t* d_words, float fill, float* d_sdf)
{
    ProfScope p(ctx, d_sdf ? VP_K_JFA_LAST : k * 4 >= f.n ? VP_K_JFA_SPARSE : VP_K_JFA_DENSE);
    if (d_sdf) {
        if (win_compact(f.n)) return launch_dense_idc_last(ctx, f, k, in, out, stride, d_words, fill, d_sdf);
        if (f.n <= 512)       return launch_dense_id9_last(ctx, f, k, in, out, stride, d_words, fill, d_sdf);
        return launch_dense_id10_last(ctx, f, k, in, out, stride, d_words, fill, d_sdf);
    }
    if (win_compact(f.n)) return launch_dense_idc_pass(ctx, f, k, in, out, stride, d_words, fill, d_sdf);
    if (f.n <= 512)       return launch_dense_id9_pass(ctx, f, k, in, out, stride, d_words, fill, d_sdf);
    return launch_dense_id10_pass(ctx, f, k, in, out, stride, d_words, fill, d_sdf);
}

// Passes of the halving sequence whose step is a multiple of `ranks`, counted from the first: those can run on planes dealt cyclically
// (plane z on rank z mod ranks), every plane finding its z -+ k on its own rank.  The first two are the fused start, so fewer than two
// (or a grid the tile kernels do not serve, or a rank count that is not a power of two dividing n into multiples of 8 planes) is 0.
uint32_t jfa_cyclic_passes(uint32_t n, uint32_t ranks)
{
    if (n < kTileMinN || ranks < 2 || (ranks & (ranks - 1)) != 0 || n % ranks != 0 || (n / ranks) % 8 != 0) return 0;
    uint32_t c = 0;
    for (uint32_t k = n / 2; k >= 1 && k % ranks == 0; k /= 2) ++c;
    return c >= 2 ? c : 0;
}

int launch_win_pass_cyclic(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t ranks, uint32_t rank)
{
    ProfScope p(ctx, VP_K_JFA_DENSE);
    if (win_compact(f.n)) return launch_cyclic_idc_pass(ctx, f, k, in, out, ranks, rank);
    if (f.n <= 512)       return launch_cyclic_id9_pass(ctx, f, k, in, out, ranks, rank);
    return launch_cyclic_id10_pass(ctx, f, k, in, out, ranks, rank);
}

bool jfa_pass_can_fuse_final(const Frame& f, uint32_t k, int algo)
{
    (void)k;
    return algo == VP_ALGO_TILED && f.n >= kTileMinN;
}

// The slab passes on PLAIN ids in caller-addressed planes (vp_jfa_pass / vp_jfa_last_pass).  VP_ALGO_NAIVE: the direct kernel, any
// buffers.  VP_ALGO_TILED: the table kernel below n = 96; the tile kernel where the three buffers are one run of consecutive planes
// (whole grids, and slabs whose halo planes lie right below / above them) and the ids are 4 bytes wide; anything else has to come as a
// window (vp_jfa_window_*) -- VP_ERR_UNSUPPORTED.
// d_sdf != nullptr: this is the last pass and it writes the sdf directly (only where jfa_pass_can_fuse_final() says so).
int launch_jfa_pass_ex(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                       const void* d_plus, void* d_out, int algo, const uint32_t* d_words, float fill, float* d_sdf)
{
    const uint32_t nz = f.z1 - f.z0;
    // the direct kernel (one thread per voxel, any buffers); as the last pass it is followed by the id -> sdf conversion of its output
    auto direct = [&]() -> int {
        {
            ProfScope p(ctx, VP_K_JFA_PASS);
            const dim3 blocks(f.n * f.n / 256, nz);
            if (wide(f))
                hipLaunchKernelGGL(jfa_pass_direct<Id64>, blocks, dim3(256), 0, ctx->stream, f, k, (const uint2*)d_in,
                                   (const uint2*)d_minus, (const uint2*)d_plus, (uint2*)d_out);
            else if (f.n <= 512)
                hipLaunchKernelGGL(jfa_pass_direct<Id9>, blocks, dim3(256), 0, ctx->stream, f, k, (const uint32_t*)d_in,
                                   (const uint32_t*)d_minus, (const uint32_t*)d_plus, (uint32_t*)d_out);
            else
                hipLaunchKernelGGL(jfa_pass_direct<Id10>, blocks, dim3(256), 0, ctx->stream, f, k, (const uint32_t*)d_in,
                                   (const uint32_t*)d_minus, (const uint32_t*)d_plus, (uint32_t*)d_out);
            VP_HIP(hipGetLastError());
        }
        return d_sdf ? launch_jfa_final(ctx, f, d_words, d_out, fill, d_sdf) : 0;
    };
    if (algo == VP_ALGO_NAIVE) return direct();
    if (f.n < kTileMinN) {
        {
            ProfScope p(ctx, VP_K_JFA_PASS);
            const int RY = (int)(256 / f.n);
            const dim3 grid((f.n + RY - 1) / RY, nz);
            const size_t lds = (size_t)(2 + RY) * kTableKernelTab * sizeof(float);
            hipLaunchKernelGGL(jfa_pass_table, grid, dim3(256), lds, ctx->stream, f, k, (const uint32_t*)d_in, (const uint32_t*)d_minus,
                               (const uint32_t*)d_plus, (uint32_t*)d_out, RY);
            VP_HIP(hipGetLastError());
        }
        return d_sdf ? launch_jfa_final(ctx, f, d_words, d_out, fill, d_sdf) : 0;
    }
    // The tile kernel runs on PLAIN ids only where they ARE a window: 4-byte ids (n <= 1024) and the three buffers one run of consecutive
    // planes.  Anything else -- 8-byte ids, halo buffers of their own -- is served by the direct kernel: same ids, bit for bit (ABI v4
    // served these cases; v5 refused them).
    const size_t plane = win_plane_bytes(f.n);
    const char* in = (const char*)d_in;
    const uint32_t pbase = std::max(f.z1, f.z0 + k);
    if (wide(f) || (f.z0 > 0 && (const char*)d_minus + (size_t)k * plane != in) || (f.z1 < f.n && (const char*)d_plus != in + (size_t)(pbase - f.z0) * plane))
        return direct();
    // the same planes seen as windows that start at global plane 0 (never dereferenced outside the planes a pass reads)
    IdWin wi, wo;
    wi.base = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(d_in) - (uintptr_t)f.z0 * plane); wi.planes = f.n; wi.at = f.z0;
    wo.base = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(d_out) - (uintptr_t)f.z0 * plane); wo.planes = f.n; wo.at = f.z0;
    return launch_win_pass(ctx, f, k, wi, wo, k, d_words, fill, d_sdf);
}

int launch_jfa_pass(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                    const void* d_plus, void* d_out, int algo)
{
    return launch_jfa_pass_ex(ctx, f, k, d_in, d_minus, d_plus, d_out, algo, nullptr, 0.0f, nullptr);
}

int launch_jfa_final(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const void* d_ids, float fill, float* d_sdf)
{
    const size_t total4 = (size_t)f.n * f.n * (f.z1 - f.z0) / 4;
    const unsigned blocks = (unsigned)((total4 + 255) / 256);
    ProfScope p(ctx, VP_K_JFA_FINAL);
    if (wide(f)) hipLaunchKernelGGL(jfa_final<Id64>, dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, (const uint2*)d_ids, fill, (float4*)d_sdf);
    else if (f.n <= 512) hipLaunchKernelGGL(jfa_final<Id9>, dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, (const uint32_t*)d_ids, fill, (float4*)d_sdf);
    else         hipLaunchKernelGGL(jfa_final<Id10>, dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, (const uint32_t*)d_ids, fill, (float4*)d_sdf);
    VP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ interleave
// The re-deal of the transposed pipeline: after the all-to-all a rank holds `ranks` chunks of `count` planes, chunk s = the planes
// b0 + s, b0 + s + ranks, ... of its (widened) slab as rank s kept them; this weaves them into consecutive planes.  A plain copy at 16
// bytes per lane; one workgroup column per destination plane.
namespace {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
win_interleave(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t quadsPerPlane, uint32_t ranks, uint32_t count)
{
    const uint32_t p = blockIdx.y;                                 // destination plane j * ranks + s
    const uint32_t s = p % ranks, j = p / ranks;
    const u32x4* src = in + (size_t)(s * count + j) * quadsPerPlane;
    u32x4* dst = out + (size_t)p * quadsPerPlane;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < quadsPerPlane; i += gridDim.x * 256u)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
}  // namespace

int launch_win_interleave(vp_ctx* ctx, uint32_t n, const IdWin& in, const IdWin& out, uint32_t ranks, uint32_t count)
{
    ProfScope p(ctx, VP_K_JFA_REDEAL);
    const uint32_t planes = ranks * count;
    const uint32_t qWord = n * n / 4, qByte = n * n / 16;          // 16-byte quads per word plane / byte plane (n % 32 == 0)
    const dim3 grid(std::min(64u, (qWord + 255u) / 256u), planes);
    hipLaunchKernelGGL(win_interleave, grid, dim3(256), 0, ctx->stream, (const u32x4*)win_words(in, n, 0), (u32x4*)win_words(out, n, out.at), qWord, ranks, count);
    if (win_compact(n))
        hipLaunchKernelGGL(win_interleave, dim3(std::min(64u, (qByte + 255u) / 256u), planes), dim3(256), 0, ctx->stream,
                           (const u32x4*)win_bytes_plane(in, n, 0), (u32x4*)win_bytes_plane(out, n, out.at), qByte, ranks, count);
    VP_HIP(hipGetLastError());
    return 0;
}

// Every id of a window := "none" (vp_jfa_window_clear): what a pipeline whose regions are rounded outwards to whole tiles starts from, so
// that the planes a pass reads without needing them hold ids of the window's own format.
int launch_win_clear(vp_ctx* ctx, uint32_t n, const IdWin& w)
{
    const size_t vox = (size_t)n * n * w.planes;
    if (win_compact(n)) {
        VP_HIP(hipMemsetD32Async((hipDeviceptr_t)w.base, (int)IdC::kNoneWord, vox, ctx->stream));
        VP_HIP(hipMemsetAsync(w.base + vox * 4, (int)IdC::kNoneByte, vox, ctx->stream));
    } else {
        VP_HIP(hipMemsetD32Async((hipDeviceptr_t)w.base, (int)(n <= 512 ? kNone9 : kNone10), vox, ctx->stream));
    }
    return 0;
}

}  // namespace vp
