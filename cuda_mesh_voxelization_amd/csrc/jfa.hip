// jfa.hip -- Jump-Flooding signed squared distance field for gfx950 (MI355X).
//
// Result contract: the sdf of the reference's sequential JFA
// (/root/reference/vplib/src/jfa/sequential.cpp:7-127), bit for bit: same passes (k = n/2 .. 1,
// :72), same 26-neighbour scan order (z, y, x outer->inner, :86-88), strict '<' acceptance (:106),
// same float expressions for positions (:79-81) and distances (jfa/jfa.h:19-20), Jacobi update.
//
// State: the reference keeps float sdf + float3 seed position per voxel (16 B, two copies, plus a
// deep copy per pass, :123-124).  Here the state is ONE uint32 per voxel -- the packed voxel
// coordinates of the best seed so far (scr(y)<<2 | x<<12 | scr(z)<<22, 0xFFFFFFFF = none).  The seed position
// and the distance are recomputed from it with the reference's expressions, which gives the same
// floats because the reference's stored sdf is itself the result of exactly that expression.
// Ping-pong between two id volumes; the last step converts ids to floats.
//
// Kernels
//   jfa_init      bitmask -> ids (and/or border bitmask).  One lane = one 32-voxel word: the 26-
//                 neighbourhood test is 27 word loads + shifts/ANDs; ids leave as coalesced 16-B
//                 stores after a wave shuffle transposes word-per-lane into voxels-per-lane.
//   jfa_pass_direct   (VP_ALGO_NAIVE) one thread per voxel, everything recomputed inline.
//   jfa_pass_chain    (VP_ALGO_TILED, n >= 256) LDS coordinate tables + register sliding window over
//                 rows k apart; jfa_pass_table is the small-n variant of the table idea.
//   jfa_final     ids + bitmask -> float sdf.
//
// Built with -ffp-contract=off (an FMA changes the result).
#include "vp_internal.h"

#pragma clang fp contract(off)

namespace vp {

namespace {

// id layout: bits [2..11] scr(y), [12..21] x, [22..31] scr(z), bits [0,1] zero -- every field is already
// a byte offset into a float table after one shift+mask.  y sits in the low field because its table is
// the one looked up once per candidate AND chain step (a single AND); x and z are decoded once per id.
// kNone has bits 0,1 set.
// The y and z fields hold scr(y), scr(z): the low five bits XORed with the next five.  Seeds reached
// by jumps of 2^j >= 32 differ from the voxel only in high coordinate bits; unscrambled they would all
// index the same LDS bank of the TY/TZ tables (measured: passes k = 32, 16 ran 2x slower).  scr is an
// involution and stays inside [0, n) because n % 32 == 0.
__device__ __forceinline__ uint32_t scr(uint32_t i) { return i ^ ((i >> 5) & 31u); }
__device__ __forceinline__ uint32_t pack_id(uint32_t x, uint32_t y, uint32_t z) { return (scr(y) << 2) | (x << 12) | (scr(z) << 22); }

// jfa/sequential.cpp:79-81 / :32-34 : voxel corner position along one axis
__device__ __forceinline__ float axis_pos(float o, uint32_t i, float vs) { return o + ((float)(int)i * vs); }

// jfa/jfa.h:19-20 with p1 = seed position decoded from `id`, p0 = (px,py,pz)
__device__ __forceinline__ float seed_distance(const Frame& f, uint32_t id, float px, float py, float pz)
{
    const float sx = axis_pos(f.ox, (id >> 12) & 1023u, f.vs);
    const float sy = axis_pos(f.oy, scr((id >> 2) & 1023u), f.vs);
    const float sz = axis_pos(f.oz, scr(id >> 22), f.vs);
    return ((sx - px) * (sx - px)) + ((sy - py) * (sy - py)) + ((sz - pz) * (sz - pz));
}

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

template <bool IDS, bool MASK>
__global__ void __launch_bounds__(256)
jfa_init(Frame f, const uint32_t* __restrict__ words, const uint32_t* __restrict__ below,
         const uint32_t* __restrict__ above, uint32_t* __restrict__ ids, uint32_t* __restrict__ border_words)
{
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
    if (centre != 0u) {
        uint32_t interior = 0xFFFFFFFFu;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                const uint32_t c = (dz == 0 && dy == 0) ? centre : grid_word(f, words, below, above, xw, y + dy, zg + dz);
                const uint32_t p = grid_word(f, words, below, above, xw - 1, y + dy, zg + dz);
                const uint32_t n = grid_word(f, words, below, above, xw + 1, y + dy, zg + dz);
                const uint32_t left = (c << 1) | (p >> 31);       // bit i = voxel x-1
                const uint32_t right = (c >> 1) | (n << 31);      // bit i = voxel x+1
                interior &= left & c & right;
            }
        border = centre & ~interior;                              // sequential.cpp:28-55
    }
    if (MASK) border_words[wi] = border;
    if (IDS) {
        const uint32_t mybase = pack_id((uint32_t)xw * 32u, (uint32_t)y, (uint32_t)zg);
        const int sub = (lane & 7) * 4;
        uint4* out = reinterpret_cast<uint4*>(ids + wbase * 32);
#pragma unroll 4
        for (int j = 0; j < 8; ++j) {
            const int src = j * 8 + (lane >> 3);
            const uint32_t b = (__shfl(border, src) >> sub) & 0xFu;
            const uint32_t id0 = __shfl(mybase, src) + ((uint32_t)sub << 12);
            uint4 v;
            v.x = (b & 1u) ? id0 : kNone;
            v.y = (b & 2u) ? id0 + (1u << 12) : kNone;
            v.z = (b & 4u) ? id0 + (2u << 12) : kNone;
            v.w = (b & 8u) ? id0 + (3u << 12) : kNone;
            out[j * 64 + lane] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ pass
// Plane of global z `zg` among the three id buffers of a slab (see vphip.h, vp_jfa_pass).
__device__ __forceinline__ const uint32_t* id_plane(const Frame& f, uint32_t k, const uint32_t* in,
                                                    const uint32_t* minus, const uint32_t* plus, int zg)
{
    const size_t plane = (size_t)f.n * f.n;
    if (zg < (int)f.z0) return minus + (size_t)(zg - ((int)f.z0 - (int)k)) * plane;
    if (zg >= (int)f.z1) {
        const int pbase = max((int)f.z1, (int)f.z0 + (int)k);
        return plus + (size_t)(zg - pbase) * plane;
    }
    return in + (size_t)(zg - (int)f.z0) * plane;
}

__global__ void __launch_bounds__(256)
jfa_pass_direct(Frame f, uint32_t k, const uint32_t* __restrict__ in, const uint32_t* __restrict__ minus,
                const uint32_t* __restrict__ plus, uint32_t* __restrict__ out)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)f.n * f.n * (f.z1 - f.z0);
    if (gid >= total) return;
    const int N = (int)f.n;
    const int x = (int)(gid % f.n);
    const int y = (int)((gid / f.n) % f.n);
    const int zg = (int)(gid / ((size_t)f.n * f.n)) + (int)f.z0;
    const float px = axis_pos(f.ox, x, f.vs), py = axis_pos(f.oy, y, f.vs), pz = axis_pos(f.oz, zg, f.vs);

    uint32_t best = in[gid];
    float bestd = (best == kNone) ? INFINITY : seed_distance(f, best, px, py, pz);   // = fabs(sdf), :84
    for (int dz = -1; dz <= 1; ++dz) {
        const int nz = zg + dz * (int)k;
        if (nz < 0 || nz >= N) continue;
        const uint32_t* pl = id_plane(f, k, in, minus, plus, nz);
        for (int dy = -1; dy <= 1; ++dy) {
            const int ny = y + dy * (int)k;
            if (ny < 0 || ny >= N) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                if (dx == 0 && dy == 0 && dz == 0) continue;
                const int nx = x + dx * (int)k;
                if (nx < 0 || nx >= N) continue;
                const uint32_t c = pl[(size_t)ny * N + nx];
                if (c != kNone) {                                  // fabs(seed) < INFINITY, :102
                    const float d = seed_distance(f, c, px, py, pz);
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
// drops the 4 n^3-byte id volume jfa_init would write and this pass would read back.
// One wave = one 64-voxel x-segment.  Requires n % 128 == 0, so k is a multiple of 64 and every candidate
// segment of a wave is exactly two aligned mask words.  Lane q of the wave fetches the words of candidate
// segment q -- ONE vector load instruction per wave (the vector-memory instruction rate, not bytes, is
// what limits these kernels; one scalar load per segment was measured 10x slower: the scalar cache thrashes
// on 16 lines per wave) -- and v_readlane distributes the 27 masks as wave-uniform values, so segments
// without border bits are skipped with scalar branches.  The pass is a pure store stream: 4 n^3 bytes out.
// `border` is the border mask of the WHOLE grid (vp_surface); the kernel produces the planes of `f`.
constexpr int kFirstRows = 8;     // rows per wave: their mask loads are all in flight before the first is used

__global__ void __launch_bounds__(256)
jfa_first_pass(Frame f, uint32_t k, const uint32_t* __restrict__ border, uint32_t* __restrict__ out)
{
    const int N = (int)f.n;
    const int lane = threadIdx.x & 63;
    const int x0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 256u + (threadIdx.x & ~63u)));   // segment start
    if (x0 >= N) return;                                            // whole wave
    const int x = x0 + lane;
    const int ybase = blockIdx.y * kFirstRows;
    const int zl = blockIdx.z;
    const int zg = zl + (int)f.z0;

    // Lane q (< 27) fetches the two mask words of candidate segment q, so the whole wave issues ONE vector
    // load per row; v_readlane then hands every lane all 27 segment masks as wave-uniform values.
    const int q = lane;
    const int qz = zg + (q / 9 - 1) * (int)k, qdy = ((q / 3) % 3 - 1) * (int)k, qx0 = x0 + (q % 3 - 1) * (int)k;
    const bool qin = q < 27 && qz >= 0 && qz < N && qx0 >= 0 && qx0 < N;
    uint2 mine[kFirstRows];
#pragma unroll
    for (int r = 0; r < kFirstRows; ++r) {
        const int ny = ybase + r + qdy;
        mine[r] = make_uint2(0u, 0u);
        if (qin && ny >= 0 && ny < N)
            mine[r] = *reinterpret_cast<const uint2*>(border + ((((size_t)qz * N + ny) * N + qx0) >> 5));   // 8-byte aligned: qx0 % 64 == 0
    }
    const float px = axis_pos(f.ox, x, f.vs), pz = axis_pos(f.oz, zg, f.vs);
#pragma unroll
    for (int r = 0; r < kFirstRows; ++r) {
        const int y = ybase + r;
        const unsigned long long own = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine[r].x, 13) |
                                       ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine[r].y, 13) << 32);
        uint32_t best = kNone;
        float bestd = INFINITY;
        if ((own >> lane) & 1ull) { best = pack_id(x, y, zg); bestd = 0.0f; }            // own seed: distance 0 (:56)
        // one ballot tells whether ANY neighbour segment holds a border voxel; for most waves none does
        if (__any(lane != 13 && (mine[r].x | mine[r].y) != 0u)) {
            const float py = axis_pos(f.oy, y, f.vs);
#pragma unroll
            for (int c = 0; c < 27; ++c) {                         // reference scan order z, y, x (sequential.cpp:86-88)
                if (c == 13) continue;
                const unsigned long long m = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine[r].x, c) |
                                             ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine[r].y, c) << 32);
                if (m == 0ull) continue;                           // scalar branch
                const int dz = c / 9 - 1, dy = (c / 3) % 3 - 1, dx = c % 3 - 1;
                const bool has = (m >> lane) & 1ull;
                const uint32_t id = pack_id((uint32_t)(x + dx * (int)k), (uint32_t)(y + dy * (int)k), (uint32_t)(zg + dz * (int)k));
                const float d = seed_distance(f, id, px, py, pz);
                const bool take = has & (d < bestd);
                bestd = take ? d : bestd;
                best = take ? id : best;
            }
        }
        out[((size_t)zl * N + y) * N + x] = best;
    }
}

// Table variant.  Workgroup = RY consecutive x-rows of one z (RY = 1 when n >= 256).
// LDS: PX[i] = ox + i*vs; TZ[i] = (PZ[i]-pz)^2 for this z; TY[r][i] = (PY[i]-py_r)^2 for row r.
// dist = ((PX[ix]-px)^2 + TY[iy]) + TZ[iz]  -- the same float operations as seed_distance().
constexpr int kTab = 1024;     // table stride: any 10-bit field (also those of kNone) stays in bounds

__global__ void __launch_bounds__(256)
jfa_pass_table(Frame f, uint32_t k, const uint32_t* __restrict__ in, const uint32_t* __restrict__ minus,
               const uint32_t* __restrict__ plus, uint32_t* __restrict__ out, int RY, const uint32_t* __restrict__ zorder)
{
    extern __shared__ float lds[];
    float* PX = lds;
    float* TZ = lds + kTab;
    float* TY = lds + 2 * kTab;

    const int N = (int)f.n;
    const int tid = threadIdx.x;
    const int zl = zorder ? (int)zorder[blockIdx.y] : (int)blockIdx.y;
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

    int r, xs, xstep;
    if (N >= 256) { r = 0; xs = tid; xstep = 256; }
    else          { r = tid / N; xs = tid - r * N; xstep = N; if (r >= RY) return; }
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

    for (int x = xs; x < N; x += xstep) {
        const float px = PX[x];
        const int xm = x - (int)k, xp = x + (int)k;
        const bool hasM = xm >= 0, hasP = xp < N;

        uint32_t c[27];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const uint32_t* rw = rows[q];
            c[q * 3 + 0] = (rw && hasM) ? rw[xm] : kNone;
            c[q * 3 + 1] = rw ? rw[x] : kNone;
            c[q * 3 + 2] = (rw && hasP) ? rw[xp] : kNone;
        }
        uint32_t best = c[13];
        float bestd = INFINITY;
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            // own state first (it wins ties: acceptance is strict, sequential.cpp:106), then scan order
            const int q = (j == 0) ? 13 : (j <= 13 ? j - 1 : j);
            const uint32_t id = c[q];
            const float sx = *reinterpret_cast<const float*>(tx + ((id >> 10) & 0xFFCu));
            const float dy2 = *reinterpret_cast<const float*>(ty + (id & 0xFFCu));
            const float dz2 = *reinterpret_cast<const float*>(tz + ((id >> 20) & 0xFFCu));
            const float dxv = sx - px;
            const float d = ((dxv * dxv) + dy2) + dz2;
            const bool take = (id != kNone) && (d < bestd);
            bestd = take ? d : bestd;
            best = take ? id : best;
        }
        orow[x] = best;
    }
}

// Fast path for n >= 256 ("chain" kernel).  A workgroup owns kChain rows of one plane that are k
// apart: y_j = r + (j0 + j) * k.  Row y_j reads rows y_j - k, y_j, y_j + k, i.e. its chain
// neighbours, so a thread walks the chain with a sliding 3-row window of candidate ids in registers
// and loads only ONE new row triple (9 ids) per voxel instead of 27 -- the L1/TA path, not HBM, was
// the limiter of the straightforward version (measured: 27 loads/voxel cost 43 % of the pass).
// Per-row LDS tables at fixed addresses turn a candidate id into (seed x, dy^2, dz^2) with 5 VALU
// ops + 3 ds_read_b32.
//   SKIP = true   (early passes, k >= n/4: sparse state, many rows outside the grid) rows / planes outside
//                 the grid are skipped with wave-uniform branches and a candidate column in which no lane
//                 of the wave holds a seed is skipped after a ballot.  (A table-free per-voxel kernel and
//                 the branch-free variant were both measured slower for these passes.)
//   SKIP = false  the voxel loop is branch-free (rows outside the grid read a row of kNone through a uniform
//                 pointer select), so the scheduler overlaps the table reads of all candidates.
//   CHECK_NONE = false (n < 1024): table slot 1023 can never be a real scrambled coordinate; it
//                 holds +inf, so a kNone candidate yields d = inf/NaN and loses without a compare.
//   FINAL = true  last pass (k = 1) fused with the id -> sdf conversion of jfa_final: the winning
//                 distance is already in a register, so the pass writes floats instead of ids and the
//                 separate read+write of the id volume disappears.
constexpr int kChain = 4;

// launch_bounds: 5 waves/SIMD (<= 96 VGPRs) measured best -- 4 (99 VGPRs) is 6 % slower, 6 spills.
template <bool SKIP, bool CHECK_NONE, bool FINAL>
__global__ void __launch_bounds__(256, 5)
jfa_pass_chain(Frame f, uint32_t k, const uint32_t* __restrict__ in, const uint32_t* __restrict__ minus,
               const uint32_t* __restrict__ plus, uint32_t* __restrict__ out, const uint32_t* __restrict__ zorder,
               const uint32_t* __restrict__ none_row, const uint32_t* __restrict__ words, float fill, float* __restrict__ sdf)
{
    __shared__ float PX[kTab];
    __shared__ float TZ[kTab];
    __shared__ float TY[kChain][kTab];

    const int N = (int)f.n;
    const uint32_t tid = threadIdx.x;
    const int r = (int)(blockIdx.x % k);
    const int j0 = (int)(blockIdx.x / k) * kChain;
    const int ybase = r + j0 * (int)k;                           // row of chain element 0 of this workgroup
    const int zl = zorder ? (int)zorder[blockIdx.y] : (int)blockIdx.y;
    const int zg = zl + (int)f.z0;
    {
        const float pz = axis_pos(f.oz, zg, f.vs);
        float py[kChain];
#pragma unroll
        for (int j = 0; j < kChain; ++j) py[j] = axis_pos(f.oy, ybase + j * (int)k, f.vs);
        for (uint32_t i = tid; i < (uint32_t)N; i += 256) {
            const uint32_t si = scr(i);
            PX[i] = axis_pos(f.ox, i, f.vs);
            const float dzv = axis_pos(f.oz, i, f.vs) - pz;
            TZ[si] = dzv * dzv;
            const float sy = axis_pos(f.oy, i, f.vs);
#pragma unroll
            for (int j = 0; j < kChain; ++j) {
                const float dyv = sy - py[j];
                TY[j][si] = dyv * dyv;
            }
        }
        if (!CHECK_NONE && tid == 0) {
            PX[kTab - 1] = 0.0f; TZ[kTab - 1] = INFINITY;
#pragma unroll
            for (int j = 0; j < kChain; ++j) TY[j][kTab - 1] = 0.0f;
        }
    }
    __syncthreads();

    const char* tx = reinterpret_cast<const char*>(PX);
    const char* tz = reinterpret_cast<const char*>(TZ);
    const char* zp[3];                                             // the three source planes (byte pointers)
    bool zv[3];
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz) {
        const int nz = zg + dz * (int)k;
        zv[dz + 1] = nz >= 0 && nz < N;
        zp[dz + 1] = reinterpret_cast<const char*>(zv[dz + 1] ? id_plane(f, k, in, minus, plus, nz) : in + (size_t)zl * N * N);
    }
    const uint32_t rowBytes = (uint32_t)N * 4u;

    for (uint32_t x = tid; x < (uint32_t)N; x += 256) {
        const float px = PX[x];
        const bool hasM = x >= k, hasP = x + k < (uint32_t)N;
        const uint32_t xo = x * 4u, xmo = hasM ? xo - k * 4u : xo, xpo = hasP ? xo + k * 4u : xo;
        // 9 ids of source row yy (3 planes x {x-k, x, x+k}); kNone where the row/plane is outside the grid
        auto load_row = [&](int yy, uint32_t (&w)[9]) {
            const bool yin = yy >= 0 && yy < N;                    // wave-uniform
            const uint32_t ro = (uint32_t)(yin ? yy : 0) * rowBytes;
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                // uniform base + 32-bit lane offsets -> global_load ... saddr
                if (!SKIP) {
                    // branch-free, a row outside the grid reads from a row of kNone (uniform pointer select)
                    const char* b = (yin && zv[dz]) ? zp[dz] + ro : reinterpret_cast<const char*>(none_row);
                    w[dz * 3 + 0] = *reinterpret_cast<const uint32_t*>(b + xmo);
                    w[dz * 3 + 1] = *reinterpret_cast<const uint32_t*>(b + xo);
                    w[dz * 3 + 2] = *reinterpret_cast<const uint32_t*>(b + xpo);
                } else if (yin && zv[dz]) {
                    const char* b = zp[dz] + ro;
                    w[dz * 3 + 0] = *reinterpret_cast<const uint32_t*>(b + xmo);
                    w[dz * 3 + 1] = *reinterpret_cast<const uint32_t*>(b + xo);
                    w[dz * 3 + 2] = *reinterpret_cast<const uint32_t*>(b + xpo);
                } else {
                    w[dz * 3 + 0] = kNone; w[dz * 3 + 1] = kNone; w[dz * 3 + 2] = kNone;
                }
            }
        };

        // one chain step: window (wm, w0, wp) = rows (y-k, y, y+k); TY table of this row at `ty`
        auto step = [&](int y, const char* ty, const uint32_t (&wm)[9], const uint32_t (&w0)[9], const uint32_t (&wp)[9]) {
            uint32_t best = w0[4];
            float bestd = INFINITY;
            auto eval = [&](uint32_t id, bool ok) {
#ifdef VP_EXP_NOLDS         /* timing experiment only: same VALU work, no table reads */
                const float sx = __uint_as_float((id & 0xFFCu) | 0x3f800000u);
                const float dy2 = __uint_as_float(((id >> 10) & 0xFFCu) | 0x3f800000u);
                const float dz2 = __uint_as_float(((id >> 20) & 0xFFCu) | 0x3f800000u);
#else
                const float sx = *reinterpret_cast<const float*>(tx + ((id >> 10) & 0xFFCu));
                const float dy2 = *reinterpret_cast<const float*>(ty + (id & 0xFFCu));
                const float dz2 = *reinterpret_cast<const float*>(tz + ((id >> 20) & 0xFFCu));
#endif
                const float dxv = sx - px;
                const float d = ((dxv * dxv) + dy2) + dz2;
                bool take = ok & (d < bestd);
                if (CHECK_NONE) take = take & (id != kNone);
                bestd = take ? d : bestd;
                best = take ? id : best;
            };
            auto cand = [&](uint32_t id, bool ok) {
                if (SKIP) { if (__any(ok & (id != kNone))) eval(id, ok); }
                else eval(id, ok);
            };
            eval(w0[4], true);                       // own state first: it wins ties (strict '<', sequential.cpp:106)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {         // reference scan order: z, then y, then x (sequential.cpp:86-88)
                cand(wm[dz * 3 + 0], hasM); cand(wm[dz * 3 + 1], true); cand(wm[dz * 3 + 2], hasP);
                cand(w0[dz * 3 + 0], hasM); if (dz != 1) cand(w0[dz * 3 + 1], true); cand(w0[dz * 3 + 2], hasP);
                cand(wp[dz * 3 + 0], hasM); cand(wp[dz * 3 + 1], true); cand(wp[dz * 3 + 2], hasP);
            }
            const size_t rowIdx = (size_t)zl * N + y;
            if (FINAL) {
                // jfa_final's rule (sequential.cpp:55-60,106-109): set voxels carry +, unset ones the sign of the
                // caller's fill; bestd is +inf when no seed was found, which copysign turns into the fill itself.
                const uint32_t wbits = words[rowIdx * f.w + (x >> 5)];
                const bool set = (wbits >> (x & 31)) & 1u;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(sdf + rowIdx * N) + xo) = set ? bestd : copysignf(bestd, fill);
            } else {
                *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(out + rowIdx * N) + xo) = best;
            }
        };

        // kChain = 4 steps over a 4-row register ring: the row needed by the NEXT step is requested
        // before the current step is evaluated, so its latency hides behind ~270 VALU instructions.
        uint32_t wa[9], wb[9], wc[9], wd[9];
        const int K = (int)k;
        load_row(ybase - K, wa);
        load_row(ybase, wb);
        load_row(ybase + K, wc);
        load_row(ybase + 2 * K, wd);
        step(ybase, reinterpret_cast<const char*>(TY[0]), wa, wb, wc);
        if (ybase + K < N) {
            load_row(ybase + 3 * K, wa);
            step(ybase + K, reinterpret_cast<const char*>(TY[1]), wb, wc, wd);
            if (ybase + 2 * K < N) {
                load_row(ybase + 4 * K, wb);
                step(ybase + 2 * K, reinterpret_cast<const char*>(TY[2]), wc, wd, wa);
                if (ybase + 3 * K < N)
                    step(ybase + 3 * K, reinterpret_cast<const char*>(TY[3]), wd, wa, wb);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ final
// One lane = 4 voxels.  sequential.cpp:55-60,106-109 + apps/cli/main.cpp:200 give the sign rule.
__global__ void __launch_bounds__(256)
jfa_final(Frame f, const uint32_t* __restrict__ words, const uint4* __restrict__ ids, float fill,
          float4* __restrict__ sdf)
{
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total4 = (size_t)f.n * f.n * (f.z1 - f.z0) / 4;
    if (i4 >= total4) return;
    const size_t v = i4 * 4;
    const uint32_t x = (uint32_t)(v % f.n);
    const uint32_t y = (uint32_t)((v / f.n) % f.n);
    const uint32_t zg = (uint32_t)(v / ((size_t)f.n * f.n)) + f.z0;
    const uint32_t bits = (words[v >> 5] >> (v & 31)) & 0xFu;
    const uint4 id = ids[i4];
    const float py = axis_pos(f.oy, y, f.vs), pz = axis_pos(f.oz, zg, f.vs);
    const uint32_t idv[4] = { id.x, id.y, id.z, id.w };
    float o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const bool set = (bits >> b) & 1u;
        const float init = set ? INFINITY : fill;                  // interior +inf (:59) / caller's fill
        if (idv[b] == kNone) { o[b] = init; continue; }
        const float d = seed_distance(f, idv[b], axis_pos(f.ox, x + b, f.vs), py, pz);
        o[b] = copysignf(d, init);                                 // :108
    }
    sdf[i4] = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
int launch_jfa_init(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* below,
                    const uint32_t* above, uint32_t* d_ids, uint32_t* d_border_words)
{
    const size_t nwords = (size_t)f.n * f.n * (f.z1 - f.z0) / 32;
    const unsigned blocks = (unsigned)(nwords / 256);             // nwords is a multiple of 256
    ProfScope p(ctx, d_ids ? VP_K_JFA_INIT : VP_K_SURFACE);
    if (d_ids && d_border_words)
        hipLaunchKernelGGL((jfa_init<true, true>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, d_ids, d_border_words);
    else if (d_ids)
        hipLaunchKernelGGL((jfa_init<true, false>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, d_ids, d_border_words);
    else
        hipLaunchKernelGGL((jfa_init<false, true>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, d_ids, d_border_words);
    VP_HIP(hipGetLastError());
    return 0;
}

// Plane processing order for step k: planes z, z+k, z+2k, ... back to back, so the three planes a
// workgroup reads (z-k, z, z+k) were touched by the immediately preceding / following workgroups
// and are served from L2 / Infinity Cache instead of HBM.  Cached per (n, slab, k) on the device.
static int jfa_zorder(vp_ctx* ctx, const Frame& f, uint32_t k, const uint32_t** out)
{
    const uint32_t nz = f.z1 - f.z0;
    *out = nullptr;
    if (k < 16 || k >= nz) return 0;                               // natural order already reuses / nothing to gain
    if (ctx->zorder_n != f.n || ctx->zorder_z0 != f.z0 || ctx->zorder_z1 != f.z1) {
        std::vector<uint32_t> host;
        ctx->zorder_k.clear();
        for (uint32_t kk = f.n / 2; kk >= 1; kk /= 2) {
            if (kk < 16 || kk >= nz) continue;
            ctx->zorder_k.push_back(kk);
            for (uint32_t r = 0; r < kk; ++r)
                for (uint32_t zg = f.z0; zg < f.z1; ++zg)
                    if (zg % kk == r) host.push_back(zg - f.z0);
        }
        VP_TRY(reserve(ctx, ctx->zorder, std::max<size_t>(host.size(), 1) * 4));
        if (!host.empty()) {
            VP_HIP(hipMemcpyAsync(ctx->zorder.ptr, host.data(), host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            VP_HIP(hipStreamSynchronize(ctx->stream));
        }
        ctx->zorder_n = f.n; ctx->zorder_z0 = f.z0; ctx->zorder_z1 = f.z1;
    }
    for (size_t i = 0; i < ctx->zorder_k.size(); ++i)
        if (ctx->zorder_k[i] == k) { *out = (const uint32_t*)ctx->zorder.ptr + i * nz; return 0; }
    return 0;
}

int launch_jfa_pass(vp_ctx* ctx, const Frame& f, uint32_t k, const uint32_t* d_in, const uint32_t* d_minus,
                    const uint32_t* d_plus, uint32_t* d_out, int algo)
{
    return launch_jfa_pass_ex(ctx, f, k, d_in, d_minus, d_plus, d_out, algo, nullptr, 0.0f, nullptr);
}

bool jfa_can_start_from_mask(const Frame& f, int algo) { return algo == VP_ALGO_TILED && f.n >= 256 && f.n % 128 == 0; }

// First pass from the whole-grid border mask (see jfa_first_pass).
int launch_jfa_first_pass(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, uint32_t* d_out)
{
    ProfScope p(ctx, VP_K_JFA_PASS);
    hipLaunchKernelGGL(jfa_first_pass, dim3((f.n + 255) / 256, f.n / kFirstRows, f.z1 - f.z0), dim3(256), 0, ctx->stream, f, f.n / 2,
                       d_border, d_out);
    VP_HIP(hipGetLastError());
    return 0;
}

bool jfa_pass_can_fuse_final(const Frame& f, uint32_t k, int algo)
{
    return algo == VP_ALGO_TILED && f.n >= 256;
}

// d_sdf != nullptr: this is the last pass and it writes the sdf directly (only where
// jfa_pass_can_fuse_final() says so); otherwise ids go to d_out.
int launch_jfa_pass_ex(vp_ctx* ctx, const Frame& f, uint32_t k, const uint32_t* d_in, const uint32_t* d_minus,
                       const uint32_t* d_plus, uint32_t* d_out, int algo, const uint32_t* d_words, float fill, float* d_sdf)
{
    const size_t total = (size_t)f.n * f.n * (f.z1 - f.z0);
    const uint32_t nz = f.z1 - f.z0;
    ProfScope p(ctx, VP_K_JFA_PASS);
    if (algo == VP_ALGO_NAIVE) {
        const unsigned blocks = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(jfa_pass_direct, dim3(blocks), dim3(256), 0, ctx->stream, f, k, d_in, d_minus, d_plus, d_out);
    } else if (f.n >= 256) {
        const uint32_t* zorder = nullptr;
        VP_TRY(jfa_zorder(ctx, f, k, &zorder));
        if (!ctx->none_row.ptr) {                                  // a row of kNone for out-of-grid reads
            VP_TRY(reserve(ctx, ctx->none_row, kTab * sizeof(uint32_t)));
            VP_HIP(hipMemsetAsync(ctx->none_row.ptr, 0xFF, kTab * sizeof(uint32_t), ctx->stream));
        }
        const uint32_t* none_row = (const uint32_t*)ctx->none_row.ptr;
        const uint32_t chainLen = (f.n + k - 1) / k;               // rows per residue class
        const dim3 grid(k * ((chainLen + kChain - 1) / kChain), nz);
        const bool skip = k * 4 >= f.n, chk = f.n >= 1024, fin = d_sdf != nullptr;
#define VP_LAUNCH_CHAIN(S, C, F) hipLaunchKernelGGL((jfa_pass_chain<S, C, F>), grid, dim3(256), 0, ctx->stream, f, k, d_in, \
                                                    d_minus, d_plus, d_out, zorder, none_row, d_words, fill, d_sdf)
        if (fin)       { if (chk) VP_LAUNCH_CHAIN(false, true, true);  else VP_LAUNCH_CHAIN(false, false, true); }
        else if (skip) { if (chk) VP_LAUNCH_CHAIN(true, true, false);  else VP_LAUNCH_CHAIN(true, false, false); }
        else           { if (chk) VP_LAUNCH_CHAIN(false, true, false); else VP_LAUNCH_CHAIN(false, false, false); }
#undef VP_LAUNCH_CHAIN
    } else {
        const int RY = (int)(256 / f.n);
        const dim3 grid((f.n + RY - 1) / RY, nz);
        const size_t lds = (size_t)(2 + RY) * kTab * sizeof(float);
        hipLaunchKernelGGL(jfa_pass_table, grid, dim3(256), lds, ctx->stream, f, k, d_in, d_minus, d_plus, d_out, RY,
                           (const uint32_t*)nullptr);
    }
    VP_HIP(hipGetLastError());
    return 0;
}

int launch_jfa_final(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* d_ids,
                     float fill, float* d_sdf)
{
    const size_t total4 = (size_t)f.n * f.n * (f.z1 - f.z0) / 4;
    const unsigned blocks = (unsigned)((total4 + 255) / 256);
    ProfScope p(ctx, VP_K_JFA_FINAL);
    hipLaunchKernelGGL(jfa_final, dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, (const uint4*)d_ids, fill, (float4*)d_sdf);
    VP_HIP(hipGetLastError());
    return 0;
}

}  // namespace vp
