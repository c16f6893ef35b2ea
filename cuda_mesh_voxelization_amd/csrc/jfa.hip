// jfa.hip -- Jump-Flooding signed squared distance field for gfx950 (MI355X).
//
// Result contract: the sdf of the reference's sequential JFA
// (/root/reference/vplib/src/jfa/sequential.cpp:7-127), bit for bit: same passes (k = n/2 .. 1,
// :72), same 26-neighbour scan order (z, y, x outer->inner, :86-88), strict '<' acceptance (:106),
// same float expressions for positions (:79-81) and distances (jfa/jfa.h:19-20), Jacobi update.
//
// State: the reference keeps float sdf + float3 seed position per voxel (16 B, two copies, plus a
// deep copy per pass, :123-124).  Here the state is ONE packed id per voxel -- the voxel coordinates of
// the best seed so far (4 bytes for n <= 1024 -- two layouts, n <= 512 and above --, 8 bytes for n <= 2048).  The seed
// position and the distance are recomputed from it with the reference's expressions, which gives the
// same floats because the reference's stored sdf is itself the result of exactly that expression.
// Ping-pong between two id volumes; the last step converts ids to floats.
//
// Kernels (all templated on the id format)
//   jfa_init        bitmask -> ids (and/or border bitmask).  One lane = one 32-voxel word: the 26-
//                   neighbourhood test is 27 word loads + shifts/ANDs; ids leave as coalesced 16-B
//                   stores after a wave shuffle transposes word-per-lane into voxels-per-lane.
//   jfa_first_pass  step k = n/2 straight from the border bitmask (no init id volume).
//   jfa_pass_direct (VP_ALGO_NAIVE) one thread per voxel, everything recomputed inline.
//   jfa_pass_zstream (VP_ALGO_TILED, n >= 96) LDS coordinate tables; a workgroup owns a tile of rows x planes that
//                   are k apart, reads every source plane once and scatters each id into the outputs it is a
//                   candidate for; jfa_pass_table is the small-n variant.
//   jfa_final       ids + bitmask -> float sdf.
//
// Built with -ffp-contract=off (an FMA changes the result).
#include "vp_internal.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

// Build parts.  The tile kernel is instantiated for four id formats x pair modes x tile shapes: one translation unit took nine minutes
// of a single core.  The build compiles this file five times side by side (cuda_mesh_voxelization_amd/build.py, tools/exp_build.sh):
// part 0 holds every launcher and kernel EXCEPT the instantiations of launch_dense<ID>, parts 1 .. 4 hold one launch_dense_<format>()
// each; the kernels live in an anonymous namespace, so every part gets its own copies of what it uses and nothing else.
// VP_JFA_PART undefined (-1): everything in one unit.
#ifndef VP_JFA_PART
#define VP_JFA_PART -1
#endif
#define VP_PART_MAIN (VP_JFA_PART <= 0)
#define VP_PART_HAS(p) (VP_JFA_PART < 0 || VP_JFA_PART == (p))

namespace vp {

namespace {

// The y and z fields of an id hold scr(y), scr(z): the low five bits XORed with the next five.  Seeds
// reached by jumps of 2^j >= 32 differ from the voxel only in high coordinate bits; unscrambled they would
// all index the same LDS bank of the TY/TZ tables (measured: passes k = 32, 16 ran 2x slower).  scr is an
// involution and stays inside [0, n) because n % 32 == 0.
__device__ __forceinline__ uint32_t scr(uint32_t i) { return i ^ ((i >> 5) & 31u); }

// a / b and a % b for workgroup-uniform operands: shifts when b is a power of two (it is for every power-of-two grid), the
// ~25-instruction reciprocal sequence otherwise.  Three of these open every tile of the pass kernels.
__device__ __forceinline__ void udivmod(uint32_t a, uint32_t b, uint32_t& q, uint32_t& r)
{
    if ((b & (b - 1u)) == 0u) { q = a >> (uint32_t)__builtin_ctz(b); r = a & (b - 1u); }
    else { q = a / b; r = a % b; }
}

// Division by a run-time constant the HOST knows (Granlund - Montgomery, the branch-free 32-bit form): q = (t + ((a - t) >> s1)) >> s2 with
// t = mulhi(m, a); exact for every 32-bit a.  Four instructions instead of the ~25 of a / b: jfa_first_two divides twice per 450-instruction tile.
struct FastDiv {
    uint32_t d, m, s1, s2;
    __device__ __forceinline__ void divmod(uint32_t a, uint32_t& q, uint32_t& r) const
    {
        const uint32_t t = __umulhi(m, a);
        q = (t + ((a - t) >> s1)) >> s2;
        r = a - q * d;
    }
};
static inline FastDiv make_fastdiv(uint32_t d)
{
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < d) ++l;                         // ceil(log2 d)
    const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1ull;
    return FastDiv{d, (uint32_t)m, l < 1u ? l : 1u, l > 1u ? l - 1u : 0u};
}

// Id formats.  An accessor returns a coordinate field as a BYTE offset into a table of floats (index * 4).
//
// 32-bit formats IdU<BITS> (BITS = 9: n <= 512, BITS = 10: n <= 1024).  x sits UNSHIFTED in the low BITS + 1 bits -- its
// top bit is set only in "none", whose x index 2^BITS is therefore the first slot no real id uses: the x table has 2^BITS + 1
// entries with +inf in the last one, and "none" gets an infinite distance through the ordinary lookup at every n (no test, no
// spare-slot tricks in the y / z tables).  y and z follow, each preceded by zero guard bits where the 32 bits allow it, so
// that their byte offsets come out of ONE instruction:
//   BITS = 9 :  x [0..9] | 00 | scr(y) [12..20] | 00 | scr(z) [23..31]     yoff = bfe(id, 10, 11), zoff = id >> 21, xoff = (id & 0x3FF) << 2
//   BITS = 10:  x [0..10] | scr(y) [11..20] | 0 | scr(z) [22..31]          yoff, zoff = shift + mask, xoff = (id & 0x7FF) << 2
// (four / six decode instructions per id; round 1's layout -- three pre-shifted 10-bit fields -- needed five + a none test at n = 1024).
template <int BITS>
struct IdU {
    using T = uint32_t;
    static constexpr int kTab = 1 << BITS;        // entries of the y / z tables; the x table has kTab + 1
    static constexpr uint32_t kMask = (uint32_t)(kTab - 1) * 4u;
    static constexpr int kYS = BITS == 9 ? 12 : 11, kZS = BITS == 9 ? 23 : 22;
    static constexpr uint32_t kXMask = (2u << BITS) - 1u;
    static constexpr T kNoneValue = (1u << BITS) | ((uint32_t)(kTab - 1) << kYS) | ((uint32_t)(kTab - 1) << kZS);
    __device__ static __forceinline__ T none() { return kNoneValue; }
    __device__ static __forceinline__ bool is_none(T a) { return a == kNoneValue; }
    __device__ static __forceinline__ T pack(uint32_t x, uint32_t y, uint32_t z) { return x | (scr(y) << kYS) | (scr(z) << kZS); }
    __device__ static __forceinline__ T add_x(T a, uint32_t dx) { return a + dx; }
    __device__ static __forceinline__ uint32_t xoff(T a) { return (a & kXMask) << 2; }            // <= 4 * kTab ("none")
    __device__ static __forceinline__ uint32_t yoff(T a) { return BITS == 9 ? __builtin_amdgcn_ubfe(a, 10u, 11u) : ((a >> (kYS - 2)) & kMask); }
    __device__ static __forceinline__ uint32_t zoff(T a) { return BITS == 9 ? (a >> 21) : ((a >> (kZS - 2)) & kMask); }
    __device__ static __forceinline__ T sel(bool c, T a, T b) { return c ? a : b; }
    __device__ static __forceinline__ T join(T a, T b) { return a | b; }               // ids packed from disjoint coordinates
    __device__ static __forceinline__ T shfl(T a, int src) { return (T)__shfl((int)a, src); }
};
using Id9 = IdU<9>;
using Id10 = IdU<10>;
static_assert(Id9::kNoneValue == kNone9 && Id10::kNoneValue == kNone10, "vp_internal.h");

struct Id64 {                     // n <= 2048: .x = scr(z)<<2 | x<<13, .y = scr(y)<<2
    using T = uint2;
    static constexpr int kTab = 2048;
    static constexpr uint32_t kMask = 0x1FFCu;
    __device__ static __forceinline__ T none() { return make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu); }
    __device__ static __forceinline__ bool is_none(T a) { return a.x == 0xFFFFFFFFu; }
    __device__ static __forceinline__ T pack(uint32_t x, uint32_t y, uint32_t z) { return make_uint2((scr(z) << 2) | (x << 13), scr(y) << 2); }
    __device__ static __forceinline__ T add_x(T a, uint32_t dx) { return make_uint2(a.x + (dx << 13), a.y); }
    __device__ static __forceinline__ uint32_t zoff(T a) { return a.x & kMask; }
    __device__ static __forceinline__ uint32_t xoff(T a) { return (a.x >> 11) & kMask; }
    __device__ static __forceinline__ uint32_t yoff(T a) { return a.y & kMask; }
    __device__ static __forceinline__ T sel(bool c, T a, T b) { return make_uint2(c ? a.x : b.x, c ? a.y : b.y); }
    __device__ static __forceinline__ T join(T a, T b) { return make_uint2(a.x | b.x, a.y | b.y); }
    __device__ static __forceinline__ T shfl(T a, int src) { return make_uint2((uint32_t)__shfl((int)a.x, src), (uint32_t)__shfl((int)a.y, src)); }
};

// Compact ids (round 4; n <= 2048, whole-grid vp_jfa only): 5 bytes per voxel in TWO planes instead of the 8 of Id64 --
//   word plane (n^3 dwords):  x [0..10] | scr(y) [11..21] | low ten bits of scr(z) [22..31]
//   byte plane (n^3 bytes, right behind the word plane):  top bit of scr(z) [bit 1] | "none" [bit 2]     (values 0, 2, 4)
// 33 coordinate bits + "none" do not fit 32; the passes at this size are bound by fabric traffic, not by instruction issue
// (profiles/r03/n2048_wide_ablation.txt), so what pays is fewer bytes: 10 instead of 16 per voxel and pass.  In registers an id is a
// uint2 (.x = word, .y = byte).  The byte shifted left by 11 IS the high part of the z table offset: 2 -> slot 1024 + .., 4 ("none") ->
// slots 2048 .. 3071, which hold +inf (the tile kernel's z table has 3072 entries for this format): "none" gets an infinite distance
// through the ordinary lookup, as it does through slot TAB of the x table in IdU.
// Everything stays per lane (a dword and a byte load / store per id): no cross-lane packing of bit planes.
#ifndef VP_TILE_MIN_N
#define VP_TILE_MIN_N 96            // VP_ALGO_TILED runs the tile kernels from this side on (n % 32 == 0: 96, 128, ...); the table kernel of round 1 below it.
                                   // Rounds 1 - 3 drew the line at 256: whole step (voxelize + JFA, bunny) 0.294 -> 0.153 ms at n = 128, 0.65 -> 0.21 at 160,
                                   // 0.92 -> 0.27 at 192, 1.50 -> 0.34 ms at 224; at 96 0.148 -> 0.139; at 64 / 32 the table kernel wins (0.063 / 0.061 against
                                   // 0.117 / 0.102 ms: launch-bound) -- tools/ab_wall.py, profiles/r04/ab_tilemin.txt
#endif
#ifndef VP_IDC_NONE_Z
#define VP_IDC_NONE_Z 1             // 0: the first form of round 4 -- "none" in bit 0 of the byte, merged into the x offset (slot 2048 of the x table):
                                    // 3 more VALU per decoded id (5340 -> 4987 per 32 outputs); dense pass at n = 2048 34.8 -> 34.0 ms, step 327.2 ->
                                    // 320.7 ms (profiles/r04/ab_idcz_2048.txt: two alternating bench runs of each build on one box)
#endif
struct IdC {
    using T = uint2;
    static constexpr int kTab = 2048;
    static constexpr int kTabZ = VP_IDC_NONE_Z ? 3072 : 2048;      // entries of the z table
    static constexpr uint32_t kMask = 0x1FFCu;
    static constexpr uint32_t kNoneWord = 0xFFFFF800u;
    static constexpr uint32_t kNoneBit = VP_IDC_NONE_Z ? 4u : 1u;  // in the byte; bit 1 = top bit of scr(z)
    static constexpr uint32_t kNoneByte = VP_IDC_NONE_Z ? 4u : 3u;
    __device__ static __forceinline__ T none() { return make_uint2(kNoneWord, kNoneByte); }
    __device__ static __forceinline__ bool is_none(T a) { return (a.y & kNoneBit) != 0u; }
    __device__ static __forceinline__ T pack(uint32_t x, uint32_t y, uint32_t z)
    {
        const uint32_t sz = scr(z);
        return make_uint2(x | (scr(y) << 11) | ((sz & 1023u) << 22), (sz >> 10) << 1);
    }
    __device__ static __forceinline__ T from64(uint2 a)           // Id64 -> compact ("none" = all ones)
    {
        if (a.x == 0xFFFFFFFFu) return none();
        const uint32_t sz = (a.x >> 2) & 2047u, sy = (a.y >> 2) & 2047u;
        return make_uint2((a.x >> 13) | (sy << 11) | ((sz & 1023u) << 22), (sz >> 10) << 1);
    }
#if VP_IDC_NONE_Z
    __device__ static __forceinline__ uint32_t xoff(T a) { return (a.x & 0x7FFu) << 2; }
    __device__ static __forceinline__ uint32_t zoff(T a) { return ((a.x >> 20) & 0xFFCu) | (a.y << 11); }   // byte 2: slot 1024 + .., 4 ("none"): 2048 + ..
#else
    __device__ static __forceinline__ uint32_t xoff(T a) { return ((a.x & 0x7FFu) << 2) | ((a.y & 1u) << 13); }   // "none": 4 * 2048
    __device__ static __forceinline__ uint32_t zoff(T a) { return ((a.x >> 20) & 0xFFCu) | ((a.y & 2u) << 11); }
#endif
    __device__ static __forceinline__ uint32_t yoff(T a) { return (a.x >> 9) & kMask; }
    __device__ static __forceinline__ uint32_t zt2(T a) { return a.y & 2u; }                   // top z bit, as it sits in the byte
};

// jfa/sequential.cpp:79-81 / :32-34 : voxel corner position along one axis
__device__ __forceinline__ float axis_pos(float o, uint32_t i, float vs) { return o + ((float)(int)i * vs); }

// jfa/jfa.h:19-20 with p1 = seed position decoded from `id`, p0 = (px,py,pz)
template <class ID>
__device__ __forceinline__ float seed_distance(const Frame& f, typename ID::T id, float px, float py, float pz)
{
    const float sx = axis_pos(f.ox, ID::xoff(id) >> 2, f.vs);
    const float sy = axis_pos(f.oy, scr(ID::yoff(id) >> 2), f.vs);
    const float sz = axis_pos(f.oz, scr(ID::zoff(id) >> 2), f.vs);
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

template <class ID, bool IDS, bool MASK>
__global__ void __launch_bounds__(256)
jfa_init(Frame f, const uint32_t* __restrict__ words, const uint32_t* __restrict__ below,
         const uint32_t* __restrict__ above, typename ID::T* __restrict__ ids, uint32_t* __restrict__ border_words)
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
            store4(out, (size_t)j * 64 + lane,
                   ID::sel(b & 1u, id0, ID::none()), ID::sel(b & 2u, ID::add_x(id0, 1u), ID::none()),
                   ID::sel(b & 4u, ID::add_x(id0, 2u), ID::none()), ID::sel(b & 8u, ID::add_x(id0, 3u), ID::none()));
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
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }   // lane i <- lane i-1
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }   // lane i <- lane i+1

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
#ifndef VP_FIRST_ROWS
#define VP_FIRST_ROWS 16
#endif
constexpr int kFirstRows = VP_FIRST_ROWS;     // rows per wave: their mask loads are all in flight before the first is used

template <class ID>
__global__ void __launch_bounds__(256)
jfa_first_pass(Frame f, uint32_t k, const uint32_t* __restrict__ border, typename ID::T* __restrict__ out, uint32_t gx, uint32_t gy)
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
        out[((size_t)zl * N + y) * N + x] = best;
    }
}

constexpr int kTableKernelTab = 1024;   // entries per table of jfa_pass_table (every field offset of Id9, "none" included, stays inside)

// Table variant for small grids (n < VP_TILE_MIN_N = 96, 32-bit ids).  Workgroup = RY consecutive x-rows of one z.
// LDS: PX[i] = ox + i*vs; TZ[i] = (PZ[i]-pz)^2 for this z; TY[r][i] = (PY[i]-py_r)^2 for row r.
// dist = ((PX[ix]-px)^2 + TY[iy]) + TZ[iz]  -- the same float operations as seed_distance().
__global__ void __launch_bounds__(256)
jfa_pass_table(Frame f, uint32_t k, const uint32_t* __restrict__ in, const uint32_t* __restrict__ minus,
               const uint32_t* __restrict__ plus, uint32_t* __restrict__ out, int RY)
{
    using ID = Id9;                                                // n < 96
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

// ------------------------------------------------------------------------------------------ z-stream
// Fast path for n >= VP_TILE_MIN_N.  A workgroup owns a TILE of RY output rows x CH output planes, both k apart:
// rows y_a = y0 + a*k, planes z_j = z0 + j*k.  Output (y_a, z_j) takes its candidates from rows y_a - k, y_a,
// y_a + k of planes z_j - k, z_j, z_j + k, i.e. from the tile's own rows / planes and one halo row / plane on each
// side.  Every source plane of the tile is therefore read ONCE -- (RY+2) rows x columns {x-k, x, x+k} per thread --
// and serves up to 3 output rows x 3 output planes: 3(RY+2)(CH+2)/(RY*CH) = 6.75 (4x4) or 5.6 (4x8) loads per voxel
// instead of 27, and each row segment comes from L2 1.9 - 2.25 times instead of 9.
// The kernel is input-stationary: every id is decoded once and scattered into the running (best id, distance)
// pairs of the outputs it is a candidate for.  Planes arrive in increasing z, rows in increasing y and columns in
// increasing x, so each output still sees its 27 candidates in the reference's scan order (sequential.cpp:86-88).
// The voxel's own state no longer comes first; it wins ties instead by being merged with '<=' (a leftmost minimum
// with "own" leftmost is the same thing: candidates before it were taken with '<', later ones need '<' to replace
// it).  LDS tables at fixed addresses turn id fields into seed x, dy^2 per output row and dz^2 per output plane;
// dx^2 is computed once per id, fl(dx^2 + dy^2) once per (id, output row), which keeps the reference's association
// ((dx^2 + dy^2) + dz^2) (jfa/jfa.h:19-20).
//     per candidate-step   1 add + compare + 2 selects; the dz^2 lookup is shared by the output rows
//     per loaded id        x, y, z decode (5 VALU) + x lookup + sub, mul
//     registers            (RY+2)*3 ids of the plane in flight + as many prefetched + RY*3 running pairs: 72-76 VGPRs
// What limits it (MI355X counters, profiles/): the vector-memory pipe -- TCP busy 96 %, TD busy 85 %, 64 % of the
// time stalled on L2 returns -- while VALU and LDS sit at 60-70 %; removing all LDS lookups or 15 % of the VALU
// instructions changed nothing, fewer loads and fewer L2 requests per voxel (bigger tiles) did:
// 1x4 0.68 ms, 2x4 0.57, 3x4 0.56, 2x8 0.54, 4x4 0.52, 4x8 0.50 per dense pass at n = 512 (the sparse pass is
// fastest with 4x4: 0.35 ms; the fused last pass: 4x4 0.43, 4x8 0.41); n = 1024: 2x4 4.8 ms, 4x4 4.65, 4x8 4.93.
//   TAB           table entries.  512 for n <= 512 (2-KB tables: 18 KB of LDS per workgroup), else the id format's
//                 field range; fields are masked to it.
//   SKIP = true   (early passes, k >= n/4: sparse state, many rows outside the grid) rows / planes outside
//                 the grid are skipped with wave-uniform branches and a candidate column in which no lane
//                 of the wave holds a seed is skipped after a ballot.
//   SKIP = false  the voxel loop is branch-free: rows outside the grid read a row of "none" through a uniform
//                 base select.
//   CHECK_NONE = false (n < TAB): the last table slot can never be a real scrambled coordinate; its
//                 dz^2 entry holds +inf, so a "none" candidate yields d = inf and loses without a compare.
//   FINAL = true  last pass (k = 1) fused with the id -> sdf conversion of jfa_final: the winning
//                 distance is already in a register, so the pass writes floats instead of ids.
// The wide-id format (8-KB tables) keeps seed POSITIONS in two tables instead of squared differences per output row /
// plane (see POS below): its LDS footprint does not grow with the tile and it runs 4x8 tiles too (n = 2048: 727 -> 472 ms).
// Tried and measured slower: taking columns x-k / x+k from an LDS row buffer filled by one coalesced load per row
// (2.25 global loads per voxel, but +18 LDS operations per thread and plane and two barriers: 0.60 ms vs 0.52);
// computing the seed x from the id instead of looking it up (more VALU: 0.59); skipping the selects of a candidate
// that no lane takes (branches: 0.86); v_pk_*_f32 on pairs of ids (half rate on this part: no change); v_min_f32 for
// the distance update (0.59 vs 0.51) and an all-integer compare/select (sub, ashr, bfi, min on the bit patterns: 0.68).
constexpr int kRows = 4, kPlanes = 4;
constexpr int kPlanesDense = 8;      // dense passes with 2-KB tables (n <= 512): 4x8 tiles, 5.6 loads per voxel
constexpr int kRowsWide = 4, kPlanesWide = 8, kPlanesWideDense = 8;   // n = 2048 with position tables: 2x2 721 ms per JFA, 2x4 584, 2x8 540, 4x4 512, 4x8 472     // n = 2048: 2x2 727 ms per JFA, 2x4 726, 1x4 796, 4x2 1143

// Row loads and stores go through a buffer resource (base in SGPRs + one 32-bit VGPR byte offset that is the same
// for every row of the thread), which costs no VALU address arithmetic; plain pointer accesses from a selected base
// compiled to a 64-bit VALU add each.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_resource(const void* row, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(row), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void row_load(uint32_t& o, __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    o = __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void row_load(uint2& o, __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    o = make_uint2(v[0], v[1]);
}
__device__ __forceinline__ uint32_t row_load_u8(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ float lds_f32(const char* p) { return *reinterpret_cast<const float*>(p); }
// AUX = cache policy bits of the store (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).  Plain / sc0 / nt stores leave the line in the XCD's L2,
// sc1 forms drop it (MI355X_MICROARCH.md, "stores of each flavour").
template <int AUX = 0>
__device__ __forceinline__ void row_store(uint32_t v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)byte_off, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void row_store(float v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void row_store(uint2 v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(t, r, (int)byte_off, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void row_store_u8(uint32_t v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, r, (int)byte_off, 0, AUX);
}
// A wave-uniform value made opaque to the optimiser where it is used: the 128-bit row descriptors derived from it are
// then built right before their loads / stores (a few SALU instructions) instead of being hoisted out of the x loop,
// where the ~80 descriptors of a tile do not fit the SGPR file and were spilled to VGPR lanes (v_writelane/v_readlane).
__device__ __forceinline__ const char* opaque_uniform(const char* p)
{
    uint64_t v = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(v));
    return reinterpret_cast<const char*>(v);
}
__device__ __forceinline__ size_t opaque_uniform(size_t v)
{
    asm volatile("" : "+s"(v));
    return v;
}
// Candidate update "if (d < bestd) { bestd = d; best = id; }" as v_cmpx + plain moves under the narrowed EXEC mask:
// about 9 clocks per wave on this part against 16 for v_cmp + two v_cndmask (tools/ubench/valu_rate.hip).  The live
// mask is saved and restored INSIDE the asm (an early-clobber SGPR pair), so the statement is correct wherever the compiler
// places it -- round 1 restored EXEC from a mask captured once per x iteration, which was only right as long as every call
// stayed in the control-flow region of that capture.  Since round 2 this form only serves the fallback passes (64-bit ids,
// slabs whose halo buffers are not contiguous); the dense passes use the v_min_f64 pair update of jfa_pass_dense.
// OWN = the voxel's own state: '<=' (it wins ties).
__device__ __forceinline__ uint64_t wave_exec() { return 0; }     // kept for the call sites: the mask is no longer passed around
template <bool OWN>
__device__ __forceinline__ void take_if_closer(float& bestd, uint32_t& best, float d, uint32_t id, uint64_t)
{
    uint64_t saved;
    if (OWN)
        asm volatile("s_mov_b64 %2, exec\n\tv_cmpx_le_f32 exec, %3, %0\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\ts_mov_b64 exec, %2"
                     : "+v"(bestd), "+v"(best), "=&s"(saved) : "v"(d), "v"(id));
    else
        asm volatile("s_mov_b64 %2, exec\n\tv_cmpx_lt_f32 exec, %3, %0\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\ts_mov_b64 exec, %2"
                     : "+v"(bestd), "+v"(best), "=&s"(saved) : "v"(d), "v"(id));
}
template <bool OWN>
__device__ __forceinline__ void take_if_closer(float& bestd, uint2& best, float d, uint2 id, uint64_t)
{
    uint64_t saved;
    if (OWN)
        asm volatile("s_mov_b64 %3, exec\n\tv_cmpx_le_f32 exec, %4, %0\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\ts_mov_b64 exec, %3"
                     : "+v"(bestd), "+v"(best.x), "+v"(best.y), "=&s"(saved) : "v"(d), "v"(id.x), "v"(id.y));
    else
        asm volatile("s_mov_b64 %3, exec\n\tv_cmpx_lt_f32 exec, %4, %0\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\ts_mov_b64 exec, %3"
                     : "+v"(bestd), "+v"(best.x), "+v"(best.y), "=&s"(saved) : "v"(d), "v"(id.x), "v"(id.y));
}
// An empty asm that "modifies" a running value: everything feeding it has to be computed here.  Without it the
// compiler sinks the compare/select chains of a whole chain towards the stores and keeps every distance live
// (132 VGPRs instead of 75).
__device__ __forceinline__ void pin(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(uint32_t& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(uint2& a) { asm volatile("" : "+v"(a.x), "+v"(a.y)); }

#ifndef VP_ZSTREAM_STORE_AUX
#define VP_ZSTREAM_STORE_AUX 2      // cache policy of this kernel's output stores (row_store): 2 = nt (n = 2048: 430.9 -> 423.3 ms per step, profiles/r04)
#endif
// CZ (round 3): the tile's planes are a whole chain (CH = n / k, whole grid): the planes before the first and after the last output
// plane lie outside the grid, and their iterations are removed at compile time (see the closed tiles of jfa_pass_dense).
template <class ID, int TAB, int PXT, int RY, int CH, bool SKIP, bool CHECK_NONE, bool FINAL, bool CZ = false>
#ifndef VP_ZSTREAM_WIDE_WAVES
#define VP_ZSTREAM_WIDE_WAVES 0      // 8-byte ids: minimum waves per SIMD asked of the register allocator (0 = none: 143 VGPRs, 3 waves)
#endif
__global__ void __launch_bounds__(256, (std::is_same<ID, Id64>::value && VP_ZSTREAM_WIDE_WAVES) ? VP_ZSTREAM_WIDE_WAVES : 1)
jfa_pass_zstream(Frame f, uint32_t k, const typename ID::T* __restrict__ in, const typename ID::T* __restrict__ minus,
                 const typename ID::T* __restrict__ plus, typename ID::T* __restrict__ out,
                 const typename ID::T* __restrict__ none_row, const uint32_t* __restrict__ words, float fill, float* __restrict__ sdf, uint32_t tilesY)
{
    using T = typename ID::T;
    constexpr int kTab = TAB;                                      // table entries; fields are masked to it
    constexpr uint32_t kField = (uint32_t)(TAB - 1) * 4u;
    // 32-bit ids: PXT = TAB + 1, slots n .. TAB hold +inf -- the x index of "none" is TAB (IdU), so "none" gets an infinite
    // distance through the ordinary lookup and the offset needs no mask.  64-bit ids: PXT = TAB, fields masked to the table,
    // "none" (all ones) is caught by the spare y / z slot or a test.
    constexpr bool kU = !std::is_same<ID, Id64>::value;
    constexpr uint32_t kFieldX = (uint32_t)(PXT - 1) * 4u;
    static_assert(!kU || PXT == TAB + 1, "x table of the 32-bit formats");
    constexpr int NR = RY + 2;                                     // source rows of a plane: ybase - k .. ybase + RY*k
    constexpr int NI = NR * 3;                                     // ids per thread and plane
    __shared__ float PX[PXT];
    // POS (wide ids, 8-KB tables): TY[0] / TZ[0] hold the seed POSITIONS along y / z (scrambled index) and the squares
    // are formed per use (+2 VALU), which makes the LDS footprint independent of the tile: 24 KB instead of 8(1+RY+CH).
    constexpr bool POS = std::is_same<ID, Id64>::value;            // (no gain for the 4-KB tables of n = 1024: 36.4 vs 36.2 ms)
    __shared__ float TY[POS ? 1 : RY][kTab];
    __shared__ float TZ[POS ? 1 : CH][kTab];
    // FINAL: the bitmask words of the tile's output rows, fetched with one coalesced load while the tables are built
    // (a load at store time waits behind the prefetched ids of the next plane: +0.13 ms at n = 512)
    __shared__ uint32_t WM[FINAL ? RY * CH * (TAB / 32) : 1];

    const int N = (int)f.n;
    const int K = (int)k;
    const int nzl = (int)(f.z1 - f.z0);                            // planes in this slab
    const uint32_t tid = threadIdx.x;
    // y: residue classes of the row index mod k, RY consecutive chain elements (rows k apart) per workgroup
    const int nresY = min(K, N);
    // Every other pass walks the tiles in reverse dispatch order, so that a pass starts on the part of the volume the
    // previous pass wrote last (still in L2 / Infinity Cache).  (An XCD-aware remap of the tile index measured no gain:
    // round 1 at n = 512, round 2 at n = 2048 with 8-byte ids, 45.44 vs 45.50 ms at k = 4, 46.1 vs 47.4 ms at k = 32.)
    // (bench step 4.52 -> 4.45 ms.)
    const bool rev = ((31 - __builtin_clz(k)) & 1) != 0;
    // One-dimensional launch (round 3; a 2-D grid of the same tiles in the same order ran the passes N/2 + N/4 8 - 14 % slower,
    // see jfa_first_two): tile = y tile fastest, then z tile.
#ifndef VP_ZSTREAM_1D
#define VP_ZSTREAM_1D 1
#endif
    const uint32_t lin = rev ? gridDim.x * gridDim.y - 1 - (blockIdx.x + gridDim.x * blockIdx.y) : blockIdx.x + gridDim.x * blockIdx.y;
    uint32_t bx, by, bxq, bxr, byq, byr;
    udivmod(lin, tilesY, by, bx); udivmod(bx, (uint32_t)nresY, bxq, bxr);
    const int ybase = (int)bxr + (int)bxq * RY * K;
    // z: the same over the local plane index of the slab
    const int nres = min(K, nzl);
    udivmod(by, (uint32_t)nres, byq, byr);
    const int lbase = (int)byr + (int)byq * CH * K;
    if (ybase >= N || lbase >= nzl) return;
    const int zbase = lbase + (int)f.z0;                           // global plane of chain element 0
    float py[RY], pz[CH];                                          // positions of the output rows / planes (wave-uniform)
#pragma unroll
    for (int j = 0; j < RY; ++j) py[j] = axis_pos(f.oy, ybase + j * K, f.vs);
#pragma unroll
    for (int j = 0; j < CH; ++j) pz[j] = axis_pos(f.oz, zbase + j * K, f.vs);
    {
        for (uint32_t i = tid; i < (uint32_t)N; i += 256) {
            const uint32_t si = scr(i);
            PX[i] = axis_pos(f.ox, i, f.vs);
            const float sy = axis_pos(f.oy, i, f.vs);
            const float sz = axis_pos(f.oz, i, f.vs);
            if (POS) {
                TY[0][si] = sy;
                TZ[0][si] = sz;
            } else {
#pragma unroll
                for (int j = 0; j < RY; ++j) {
                    const float dyv = sy - py[j];
                    TY[POS ? 0 : j][si] = dyv * dyv;
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float dzv = sz - pz[j];
                    TZ[POS ? 0 : j][si] = dzv * dzv;
                }
            }
        }
        if (FINAL) {
            for (uint32_t i = tid; i < (uint32_t)(RY * CH) * f.w; i += 256) {
                const int o = (int)(i / f.w), a = o / CH, j = o % CH;
                const int oy = ybase + a * K, oz = zbase + j * K;
                WM[o * (TAB / 32) + i % f.w] = (oy < N && oz < (int)f.z1) ? words[((size_t)(oz - (int)f.z0) * N + oy) * f.w + i % f.w] : 0u;
            }
        }
        if (PXT > TAB)                                           // n == TAB: "none" (x field 512 = TAB) gets dx = inf through its x field
            for (uint32_t i = (uint32_t)N + tid; i < (uint32_t)PXT; i += 256) PX[i] = INFINITY;
        if (!CHECK_NONE && PXT == TAB && tid == 0) {
            PX[kTab - 1] = 0.0f;
#pragma unroll
            for (int j = 0; j < (POS ? 1 : RY); ++j) TY[j][kTab - 1] = 0.0f;
#pragma unroll
            for (int j = 0; j < (POS ? 1 : CH); ++j) TZ[j][kTab - 1] = INFINITY;   // POS: an infinite position gives dz^2 = inf as well
        }
    }
    __syncthreads();

    const char* tx = reinterpret_cast<const char*>(PX);
    const char* ty = reinterpret_cast<const char*>(TY);
    const char* tz = reinterpret_cast<const char*>(TZ);
    const uint32_t rowBytes = (uint32_t)N * (uint32_t)sizeof(T);
    // Output rows / planes of this workgroup that exist.  A source row or plane is read only if one of its outputs
    // exists, so nothing is fetched past the grid or past the halo of a slab.
    int yout = 1, nout = 1;
#pragma unroll
    for (int j = 1; j < RY; ++j) yout += (ybase + j * K < N) ? 1 : 0;
#pragma unroll
    for (int j = 1; j < CH; ++j) nout += (zbase + j * K < (int)f.z1) ? 1 : 0;
    const uint32_t kb = k * (uint32_t)sizeof(T);
#ifndef VP_ZSTREAM_OPAQUE
#define VP_ZSTREAM_OPAQUE (TAB <= 512)      // measured (tools/ab_step.py): sparse -7.5 %, last -2 % at n = 512; +0.5 .. 1 % with the 4-KB tables
#endif
    const int zbase0 = zbase, ybase0 = ybase;

    for (uint32_t x = tid; x < (uint32_t)N; x += 256) {
        // the uniform bases are re-read through an empty asm per x iteration, as in jfa_pass_dense: otherwise the addresses of
        // all planes of the tile are hoisted out of the x loop and spilled to VGPR lanes
        const int zbase = VP_ZSTREAM_OPAQUE ? (int)opaque_uniform((size_t)(uint32_t)zbase0) : zbase0;
        const int ybase = VP_ZSTREAM_OPAQUE ? (int)opaque_uniform((size_t)(uint32_t)ybase0) : ybase0;
        uint32_t ro[NR];                                           // byte offsets of the source rows inside a plane
        bool yv[NR];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int ny = ybase + (rr - 1) * K;
            yv[rr] = ny >= 0 && ny < N && max(rr - 2, 0) < yout;   // row rr serves output rows rr-2 .. rr
            ro[rr] = (uint32_t)(yv[rr] ? ny : 0) * rowBytes;
        }
        const float px = PX[x];
        const uint64_t ex = wave_exec();                           // EXEC of this iteration (the last one may be partial)
        const bool hasM = x >= k, hasP = x + k < (uint32_t)N;
        // A column outside the grid reads the centre column instead: the same id as the neighbouring candidate in
        // scan order, which cannot change the winner, so no validity mask is needed.
        const uint32_t xo = x * (uint32_t)sizeof(T), xmo = hasM ? xo - kb : xo, xpo = hasP ? xo + kb : xo;

        // ids of source plane zg: rows ybase-k .. ybase+RY*k x columns {x-k, x, x+k}; "none" where outside the grid or not needed
        auto load_plane = [&](int zg, T (&w)[NI], bool needed) {
            const bool zin = needed && zg >= 0 && zg < N;          // wave-uniform
            const char* pl = opaque_uniform(reinterpret_cast<const char*>(zin ? id_plane(f, k, in, minus, plus, zg) : in));
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                if (!SKIP || (zin && yv[rr])) {
                    const __amdgpu_buffer_rsrc_t b =
                        row_resource((zin && yv[rr]) ? pl + ro[rr] : reinterpret_cast<const char*>(none_row), rowBytes);
                    row_load(w[rr * 3 + 0], b, xmo);
                    row_load(w[rr * 3 + 1], b, xo);
                    row_load(w[rr * 3 + 2], b, xpo);
                } else {
                    w[rr * 3 + 0] = ID::none(); w[rr * 3 + 1] = ID::none(); w[rr * 3 + 2] = ID::none();
                }
            }
        };

        T best[RY][CH];                                            // output plane j is live from plane j-1 to plane j+1 only
        float bestd[RY][CH];

        // plane P of the chain (global plane zbase + P*k) scattered into output planes P-1, P, P+1 of every output row
        auto scatter = [&](int P, const T (&w)[NI]) {
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const int rr = q / 3, c = q % 3;
                const T id = w[q];
                auto body = [&]() {
                    const float sx = lds_f32(tx + (kU ? ID::xoff(id) : (ID::xoff(id) & kFieldX)));
                    const float dxv = sx - px;
                    // FINAL keeps distances only: "none" is given an infinite distance here, once per id
                    // n == table size has no spare +inf table slot for "none": the dense variants give it an infinite distance
                    // here, once per id (it then loses every '<'; as the own voxel it can only replace another "none"); the
                    // sparse variant, where most ids are "none", is faster with the flag (0.355 vs 0.377 ms)
                    constexpr bool kInfNone = CHECK_NONE && !SKIP;
                    const float dx2 = (kInfNone && ID::is_none(id)) ? INFINITY : dxv * dxv;
                    const uint32_t yo = ID::yoff(id) & kField, zo = ID::zoff(id) & kField;
                    const bool real = (CHECK_NONE && !kInfNone) ? !ID::is_none(id) : true;
                    const float sy = POS ? lds_f32(ty + yo) : 0.0f, sz = POS ? lds_f32(tz + zo) : 0.0f;
                    float dz2v[3];                                 // dz^2 to output planes P-1, P, P+1: shared by the output rows
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int o = P - 1 + t;
                        if (o < 0 || o >= CH) continue;
                        if (POS) { const float dzv = sz - pz[o]; dz2v[t] = dzv * dzv; }
                        else dz2v[t] = lds_f32(tz + o * (kTab * 4) + zo);
                    }
#pragma unroll
                    for (int a = rr - 2; a <= rr; ++a) {           // output rows this source row is a candidate for
                        if (a < 0 || a >= RY) continue;
                        float dy2;
                        if (POS) { const float dyv = sy - py[a]; dy2 = dyv * dyv; }
                        else dy2 = lds_f32(ty + a * (kTab * 4) + yo);
                        const float pre = dx2 + dy2;
                        const bool ownRow = (rr == a + 1) && (c == 1);
#pragma unroll
                        for (int o = P - 1; o <= P + 1; ++o) {
                            if (o < 0 || o >= CH) continue;
                            const float d = pre + dz2v[o - (P - 1)];
                            // strict '<' of sequential.cpp:106; the voxel's own state wins ties (see header)
                            if (FINAL) {
                                // the last pass only needs the winning distance: a plain minimum (no NaNs can occur), 1 VALU
                                // instead of compare + 2 selects; which candidate wins a tie no longer matters
                                float nb;
                                asm("v_min_f32 %0, %1, %2" : "=v"(nb) : "v"(bestd[a][o]), "v"(d));
                                bestd[a][o] = nb;
                                continue;
                            }
                            if (!SKIP) {                           // dense variants: v_cmpx update (-7 %)
                                if (ownRow && o == P) take_if_closer<true>(bestd[a][o], best[a][o], d, id, ex);
                                else take_if_closer<false>(bestd[a][o], best[a][o], d, id, ex);
                                continue;
                            }
                            bool take = (ownRow && o == P) ? (d <= bestd[a][o]) : (d < bestd[a][o]);
                            if (CHECK_NONE && SKIP) take = take & real;
                            bestd[a][o] = take ? d : bestd[a][o];
                            best[a][o] = ID::sel(take, id, best[a][o]);
                        }
                    }
                };
                if (SKIP) { if (__any(!ID::is_none(id))) body(); }  // a candidate column in which no lane holds a seed
                else body();
                if (!SKIP && c == 2) {                             // end of a source row: finish its updates before the next row's decode
#pragma unroll
                    for (int a = rr - 2; a <= rr; ++a)
#pragma unroll
                        for (int o = P - 1; o <= P + 1; ++o)
                            if (a >= 0 && a < RY && o >= 0 && o < CH) { pin(bestd[a][o]); pin(best[a][o]); }
                }
            }
        };
        auto store = [&](int a, int j) {
            const size_t rowIdx = opaque_uniform((size_t)(zbase + j * K - (int)f.z0) * N + (ybase + a * K));
            if (FINAL) {
                const bool set = (WM[(a * CH + j) * (TAB / 32) + (x >> 5)] >> (x & 31)) & 1u;
                // jfa_final's rule (sequential.cpp:55-60,106-109): set voxels carry +, unset ones the sign of the caller's fill;
                // bestd is +inf when no seed was found, which copysign turns into the fill itself.
                row_store<VP_ZSTREAM_STORE_AUX>(set ? bestd[a][j] : copysignf(bestd[a][j], fill), row_resource(sdf + rowIdx * N, (uint32_t)N * 4u), x * 4u);
            } else {
                row_store<VP_ZSTREAM_STORE_AUX>(best[a][j], row_resource(out + rowIdx * N, rowBytes), xo);
            }
        };

        T wa[NI], wb[NI];
        if (!CZ) load_plane(zbase - K, wa, true);
        // No branch on `nout` around the planes: a plane that is not needed reads the row of "none" (never memory past
        // the slab's halo) and its outputs are simply not stored, so the whole chain stays one basic block.
#pragma clang loop unroll(full)
        for (int P = -1; P <= CH; ++P) {
            T (&cur)[NI] = ((P + 1) & 1) ? wb : wa;                // P = -1 -> wa, 0 -> wb, ...
            T (&nxt)[NI] = ((P + 1) & 1) ? wa : wb;
            if (P + 1 <= CH - (CZ ? 1 : 0)) load_plane(zbase + (P + 1) * K, nxt, P + 1 <= nout);   // in flight while P is evaluated
            if (P + 1 < CH) {                                      // plane P is the first candidate plane of output P + 1
#pragma unroll
                for (int a = 0; a < RY; ++a) { best[a][P + 1] = ID::none(); bestd[a][P + 1] = INFINITY; }
            }
            if (!(CZ && (P == -1 || P == CH))) scatter(P, cur);
#pragma unroll
            for (int a = 0; a < RY; ++a) {
                if (P >= 1 && P - 1 < nout && a < yout) store(a, P - 1);
#pragma unroll
                for (int o = P; o <= P + 1; ++o)
                    if (o >= 0 && o < CH) { pin(bestd[a][o]); pin(best[a][o]); }
            }
#ifdef VP_EXP_FENCE
            asm volatile("" ::: "memory");
#endif
        }
    }
}

// ------------------------------------------------------------------------------------------ dense tile kernel
// Dense passes (k < n/4) and the fused last pass for 32-bit ids: the tile / stream structure of jfa_pass_zstream with a
// different candidate update and a different table layout.  What round 1's counters said about that kernel: it is bound
// by VALU issue (170 lane-instructions per voxel, 27 x (add + v_cmpx + 2 moves) of them the candidate updates) with
// the LDS pipe at 55 - 68 % behind it.
//
//  * Candidate update = ONE v_min_f64.  The running best of an output is the 64-bit pair (hi = bits of the distance,
//    lo = rank of the candidate).  A non-negative float's bit pattern orders like an unsigned integer, and a bit
//    pattern with hi <= 0x7F800000 is a finite, non-negative double whose order is that of the 64-bit pattern, so
//    v_min_f64 on such pairs IS the lexicographic minimum of (distance, rank) -- verified bit for bit on the part,
//    denormal range included (tools/ubench/probe.hip) -- at 4.3 clocks per wave against 9 for v_cmpx + 2 v_mov + the
//    EXEC restore.  The rank makes the minimum the reference's "first minimum in scan order, own state first"
//    (jfa/sequential.cpp:84-112): the own voxel has rank 0, every other candidate the byte offset of its SOURCE voxel in
//    the id volume + 1, which increases along the scan order z, y, x.  v_add_f32 writes the distance straight into the
//    high half of the candidate pair (the low half is set once per loaded id): a candidate-step is 2 VALU, 6.3 clocks,
//    instead of 4 VALU + 1 SALU, 11.6 clocks.
//    When an output is complete the seed id of its winner is fetched from where the winner was read: one gather load
//    per voxel (an L2 hit: the tile has just streamed through those lines), issued a plane ahead of its store.
//  * Tables with wide entries: TY[i] holds the squared y differences of seed coordinate i to ALL output rows of the tile
//    (RY floats = one ds_read_b128), TZ[i] those to all output planes (CH floats; the three an id needs are one
//    ds_read_b64 + one ds_read_b32).  A wide LDS read costs the same LDS cycles as a narrow one up to 8 bytes and half
//    per byte at 16 (probe.hip: b32 2.2, b64 2.2, b128 4.2 clocks per CU), so an id's 7 lookups cost 8.6 - 10.8 LDS
//    clocks instead of 15.4.
//  * FINAL keeps distances only (v_min_f32) as before and shares the tables.
// Requires the three id buffers of a slab to be contiguous in memory (one volume addressed by global plane): true for
// whole grids and for the ghost-plane slabs; other callers take jfa_pass_zstream.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double min_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));      // not fmin(): that canonicalises both operands first
    return r;
}
__device__ __forceinline__ float min3_f32(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float min_f32(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void pin(double& a) { asm volatile("" : "+v"(a)); }

// Table of W floats per seed coordinate, stored as W / E sub-tables of TAB entries of E floats (E = 1, 2 or 4): float j of
// coordinate index i lives at base + (j / E) * TAB * 4E + i * 4E + (j % E) * 4; `off` = i * 4E.  Reads floats [lo, hi]
// (compile-time constants once the plane and row loops are unrolled) with aligned b64 reads per pair (a lone float of a
// pair table is still read as the pair: a b32 on 8-byte entries would use every other bank only) and one b128 where three
// or four floats of an E = 4 entry are needed.  Fixed trip counts, so that everything folds after unrolling.
template <int W, int E, int TAB>
__device__ __forceinline__ void lds_span(const char* base, uint32_t off, int lo, int hi, float (&v)[W])
{
    static_assert(E == 1 || E == 2 || E == 4, "entry width");
#pragma unroll
    for (int g = 0; g < W / E; ++g) {
        const char* p = base + g * (TAB * 4 * E) + off;
        const int j0 = g * E;
        if (E == 1) {
            if (lo <= j0 && j0 <= hi) v[j0] = *reinterpret_cast<const float*>(p);
            continue;
        }
        if (E == 4) {
            int need = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) need += (lo <= j0 + e && j0 + e <= hi) ? 1 : 0;
            if (need >= 3) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 t = *reinterpret_cast<const f32x4*>(p);
                asm volatile("" : "+v"(t));                         // all four live: a b128 (4 LDS clocks), not the b96 (8) it is narrowed to
#pragma unroll
                for (int e = 0; e < 4; ++e) v[j0 + e] = t[e];
                continue;
            }
        }
#pragma unroll
        for (int h = 0; h < E / 2; ++h) {
            const int o0 = j0 + 2 * h, o1 = o0 + 1;
            const bool n0 = lo <= o0 && o0 <= hi, n1 = lo <= o1 && o1 <= hi;
            if (n0 || n1) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 t = *reinterpret_cast<const f32x2*>(p + h * 8);
                asm volatile("" : "+v"(t));
                v[o0] = t[0]; v[o1] = t[1];
            }
        }
    }
}

// occupancy the register allocation aims at: six waves per SIMD where the LDS footprint allows six workgroups (2-KB tables)
// or three 512-thread ones (4-KB tables); the 256-thread variant with 4-KB tables is LDS-limited to four
// FINAL form: the bitmask words of the output rows (the sign of the sdf) come from LDS (staged with the tables: 2 / 4 KB more,
// one workgroup per CU less) or straight from global memory.  Measured in the whole JFA (profiles/r02/ab22.txt): global is
// +3.4 % at n = 512 (5 -> 6 workgroups per CU does not pay for the loads) and -4.7 % at n = 1024 (2 -> 3 workgroups of 512).
#ifndef VP_FINAL_GLOBAL_MASK
#define VP_FINAL_GLOBAL_MASK 2      // 0: LDS, 1: global, 2: global with the 4-KB tables only
#endif
template <class ID> constexpr bool final_mask_global() { return VP_FINAL_GLOBAL_MASK == 1 || (VP_FINAL_GLOBAL_MASK == 2 && ID::kTab > 512); }
// 8-byte ids (n <= 2048, round 3): the same kernel with
//   * ranks relative to the tile: (index of the source row among the tile's (CH + 2) x (RY + 2) source rows) << 14 | byte offset in the row, + 1 -- the byte
//     offset of a source voxel no longer fits 32 bits (the volume is 64 GiB); a 60-entry LDS table turns the row index of the winner
//     back into its row number for the gather;
//   * 8-KB tables: PX and the squared y differences per output row (TY) as before, but ONE table of seed z POSITIONS instead of CH
//     tables of squared z differences -- (sz - pz)^2 is formed per id and output plane (2 VALU each) -- so that the footprint is
//     48 KB and three 512-thread workgroups share a CU; "none" (all ones: its fields are real coordinates at n = 2048) gets an
//     infinite seed x by an explicit test, once per id.
template <class ID> constexpr bool dense_wide() { return std::is_same<ID, Id64>::value || std::is_same<ID, IdC>::value; }   // 8-KB tables
// Compact ids (IdC, round 4): the form above with the id state in a word plane and a byte plane (10 instead of 16 bytes per voxel and pass).
// "none" goes through the slots beyond 2048 of the z table (see IdC); the top z bit of a candidate travels in bit 1 of its rank
// (rank = (tile row index + 1) << 14 | byte offset of the column + 1 | top z bit << 1; the own voxel: top z bit << 1 alone, below every
// other rank), so the winner gather fetches one dword and the byte of the output is rebuilt from the winning pair (distance = +inf: "none").
#ifndef VP_DENSE_STORE_AUX
#define VP_DENSE_STORE_AUX 2        // cache policy of the tile kernel's output stores (see row_store): nt -- the output is not read again before
                                    // the next pass; reads -15 % / -17 % (n = 512 / 1024: fewer source lines evicted), dense passes -1.1 % / -2.2 %,
                                    // sc1 forms: the same bytes, -0.4 % / -0.8 % (profiles/r04/ab_store_*.txt, pmc_bytes_store_policy_and_gather.txt)
#endif
#ifndef VP_DENSE_GATHER_NT
#define VP_DENSE_GATHER_NT 0        // winner gather with the nt policy (a line fetched for one dword should not displace halo rows): measured
                                    // +15 % / +8 % on the dense passes (n = 512 / 1024, profiles/r04/ab_gnt_ry8_*.txt): nt loads bypass the L1, where
                                    // the gathers of neighbouring lanes and rows do hit
#endif
#ifndef VP_DENSE_XCD_MAP
#define VP_DENSE_XCD_MAP 2          // 0: dispatch order; 1: XCD-contiguous tile ranges for every k; 2: only for k < 8 (see the kernel)
#endif
// Pair mode (PM = 1, 2, 4, 8; round 3): the lanes of a wave are paired so that the voxels x and x + k sit in two lanes one DPP
// permutation apart (quad_perm for k = 1, 2; row_half_mirror for k = 4; row_ror:8 for k >= 8 -- the map from lane to x below keeps
// the 64 voxels of a wave inside at most two 128-byte runs).  Each lane then loads and decodes TWO ids per source row instead of
// three -- its own column and the column beyond it (x - k for the lower lane of a pair, x + k for the upper) -- and evaluates the
// third column, which IS its partner's own column, from the partner's decoded values: seed x, squared y / z differences arrive as
// the DPP operand of the very v_sub_f32 / v_add_f32 that consumes them.  3.75 instead of 5.6 loads, decodes and table lookups per
// voxel; the candidates, their order-defining ranks and every float operation are unchanged.  A DPP operand costs the instruction
// 1.7 clocks more (tools/ubench/probe3.hip), which eats the VALU saved by the decodes: what is gained is LDS time, so the mode pays
// where the LDS pipe is the limiter -- the fused last pass -- and is left off elsewhere (launch_dense).
// Needs n a power of two and n % NT == 0 (every lane of every wave has its partner).
template <int PM>
__device__ __forceinline__ float from_partner(float v)
{
    constexpr int ctrl = PM == 1 ? 0xB1 : PM == 2 ? 0x4E : PM == 4 ? 0x141 : 0x128;   // quad_perm:[1,0,3,2] / [2,3,0,1] / row_half_mirror / row_ror:8
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}
// Closed tiles (CLOSED; round 3): at k = n/8 a chain of rows or planes has exactly eight members, so a tile of 8 rows x 8 planes IS a
// pair of whole chains: it has no halo rows and no halo planes at all -- where the 4 x 8 tile read 6 x 10 row-planes (two planes and
// one row of them outside the grid: a third of that pass's candidate steps were spent on "none") it reads 8 x 8, every one of them
// its own.  2 instead of 3.75 decoded ids per voxel (pair mode), 22.7 instead of 27 candidate steps (edge outputs have fewer
// neighbours), at the price of 48 running-pair registers (four waves per SIMD).  Compile-time: the loops simply lose their halo
// iterations.  Whole grids with n = 8 k only.  CLOSED is a mask: 1 = the rows of a tile are a whole chain (RY = n / k), 2 = the planes
// are (CH = n / k).  With the 4-KB tables (n = 1024) the 8 x 8 tile takes 68 KB of LDS -- more than the 64 KB of older parts, fine on
// gfx950 (160 KB per CU, two 512-thread workgroups = four waves per SIMD, what the 109 VGPRs allow anyway): k = 128 at n = 1024
// 2.83 -> 2.66 ms against the planes-only form (4 x 8 tiles, CLOSED = 2) that round 3 first shipped (profiles/r03/ab_clbig_1024.txt).
// FULL (round 4): every tile of the launch has all its RY output rows and CH output planes (n and the slab are multiples of RY k and
// CH k -- any whole power-of-two grid): the row / plane counts of a tile become compile-time constants and the ~220 workgroup-uniform
// branches around the stores and gathers of an x iteration (one per output row and plane, `if (a >= yout)`) disappear.
template <class ID, int RY, int CH, int NT, bool FINAL, bool ROLL, bool SKIP, int PM, int CLOSED = 0, bool FULL = false>
#ifndef VP_DENSE_WIDE_WAVES
#define VP_DENSE_WIDE_WAVES 4
#endif
#ifndef VP_DENSE_IDC_WAVES
#define VP_DENSE_IDC_WAVES 4      // compact ids, 512 threads: 6 (three workgroups per CU fit the LDS) leaves 80 VGPRs -- the fused last pass fits (78) and
                                   // gains nothing (31.0 against 30.9 - 32.1 ms), the id passes spill 50 registers: 34.0 -> 46.2 ms (gpurun r04z)
#endif
#ifndef VP_DENSE_FULL_1024
#define VP_DENSE_FULL_1024 0        // FULL id passes with the 4-KB tables under a four-wave bound (see launch_dense): -0.3 %, not adopted
#endif
__global__ void __launch_bounds__(NT, dense_wide<ID>() ? (NT == 256 ? VP_DENSE_WIDE_WAVES : (std::is_same<ID, IdC>::value ? VP_DENSE_IDC_WAVES : 4)) : RY > 4 ? 4 : (FINAL && !final_mask_global<ID>()) ? (ID::kTab == 512 ? 5 : 4) : (VP_DENSE_FULL_1024 && FULL && ID::kTab == 1024 && !FINAL) ? 4 : (ID::kTab == 512 || NT == 512) ? 6 : 4)
jfa_pass_dense(Frame f, uint32_t k, const typename ID::T* __restrict__ in, typename ID::T* __restrict__ out,
               const typename ID::T* __restrict__ none_row, const uint32_t* __restrict__ words, float fill, float* __restrict__ sdf,
               uint32_t tilesY, uint32_t tiles, uint32_t splitTiles)
{
    using T = typename ID::T;
    constexpr bool WIDE = dense_wide<ID>();
    constexpr bool CPT = std::is_same<ID, IdC>::value;             // word plane + byte plane (whole grids only: the byte plane follows the n^3 words)
    constexpr uint32_t IDB = CPT ? 4u : (uint32_t)sizeof(T);       // bytes per voxel in the (word) plane the offsets below refer to
    constexpr int TAB = ID::kTab;
    constexpr int PXT = (WIDE && (!CPT || VP_IDC_NONE_Z)) ? TAB : TAB + 1;   // 32-bit ids: slot TAB = the x index of "none" = +inf
    constexpr int TABZ = (CPT && VP_IDC_NONE_Z) ? IdC::kTabZ : TAB;      // compact ids: "none" indexes the z table beyond its 2048 real slots: z position +inf
    constexpr int EY = 1, EZ = 1;                                  // floats per table entry (wider entries: measured slower)
    constexpr int HY = (CLOSED & 1) ? 0 : 1;                       // halo rows on each side of the tile's output rows
    constexpr bool CZ = (CLOSED & 2) != 0;                         // no halo planes
    constexpr int NR = RY + 2 * HY;
    static_assert(!CLOSED || (!WIDE && !SKIP && !FINAL && ROLL), "closed tiles: dense 32-bit-id passes");
    constexpr int NC = PM ? 2 : 3;                                 // id columns a lane loads per source row
    constexpr int NI = NR * NC;
    static_assert(!PM || (!SKIP && (!WIDE || CPT)), "pair mode: dense passes on 32-bit or compact ids");
    constexpr int CHT = WIDE ? 1 : CH;                             // z tables: squared differences per output plane / one table of positions
    // (8-byte ids with a y table of POSITIONS too -- 24 KB of LDS, 256 threads, four workgroups per CU -- instead of RY tables of squared
    // differences -- 48 KB, 512 threads, two workgroups: measured 52.7 against 51.2 ms per pass at n = 2048, removed.)
    using B = typename std::conditional<FINAL, float, double>::type;
    __shared__ float PX[PXT];
    __shared__ __attribute__((aligned(16))) float TY[RY / EY][TAB][EY];
    __shared__ __attribute__((aligned(16))) float TZ[CHT / EZ][TABZ][EZ];
    __shared__ uint32_t RB[WIDE ? (CH + 2) * NR : 1];              // WIDE: row number (inside the id buffer) of every source row of the tile
    constexpr bool GM = final_mask_global<ID>();
    __shared__ uint32_t WM[(FINAL && !GM) ? RY * CH * (TAB / 32) : 1];

    const int N = (int)f.n;
    const int K = (int)k;
    const int nzl = (int)(f.z1 - f.z0);
    const uint32_t tid = threadIdx.x;
    const int nresY = min(K, N);
    const bool rev = ((31 - __builtin_clz(k)) & 1) != 0;          // alternate the traversal direction between passes
    // Workgroups are dealt round-robin to the 8 XCDs (dispatch index mod 8), each with its own L2, so neighbouring tiles --
    // which share halo rows and planes -- land on different L2s.  Re-mapping the dispatch index so that each XCD walks one
    // contiguous eighth of the tile sequence was measured twice (round 1: +-0; round 2, tools/ab_pass.py: -1 % .. +1 % at
    // n = 512, 0 .. -4 % at n = 1024; round 3 together with the tile order below: +1.5 % / +5 %): the halos come from the
    // Infinity Cache either way, and the pass is not traffic-bound.  Removed.
    // Units in dispatch order: whole tiles first, then the last `splitTiles` tiles as two half-row units each (x halves), so
    // that what the chip runs while it drains is made of short units (see launch_dense).
    uint32_t lin = blockIdx.x;
    const uint32_t total = tiles;
    uint32_t xpart = 0, xparts = 1;
    if (lin >= total - splitTiles) {
        const uint32_t u = lin - (total - splitTiles);
        lin = total - splitTiles + (u >> 1); xpart = u & 1u; xparts = 2;
    }
#if VP_DENSE_XCD_MAP
    // Workgroups are dealt to the 8 XCDs by dispatch index mod 8.  Tiles that share halo ROWS are k apart in the tile sequence (y residue
    // fastest): for k >= 8 they land on one XCD (one L2) anyway, for k = 4, 2, 1 they do not -- counters (profiles/r04/
    // pmc_bytes_*.txt, n = 1024): reads 2.8 x the id volume at k = 4 / 2 against 1.6 x at k >= 8.  For k < 8 each XCD therefore walks one
    // contiguous eighth of the whole-tile part of the sequence: reads at k = 4 / 2 fall to 1.57 x, the fused last pass 1.73 -> 1.36 x;
    // time -0.6 % / -1.8 % (the passes are not traffic-bound).  For every k (mode 1) the map LOSES at k >= 8: 2.3 x instead of 1.6 x --
    // neighbours in z are then a whole plane of tiles apart in time.
    {
        const uint32_t whole8 = (total - splitTiles) & ~7u;        // dispatch indices below total - splitTiles are whole tiles, index = tile
        if ((VP_DENSE_XCD_MAP == 1 || K < 8) && lin < whole8) lin = (lin & 7u) * (whole8 >> 3) + (lin >> 3);
    }
#endif
    if (rev) lin = total - 1u - lin;
    const int nres = min(K, nzl);
    // Tile order: y residue fastest -- consecutive tiles are adjacent rows of the volume and share nothing.  (Making the tiles of
    // one (y residue, z residue) pair -- the only ones that share halo rows and planes -- consecutive, so that the rows two
    // neighbours both read are requested at about the same time: +1.5 % / +5 %, profiles/r03/ab_order_*.txt.  Removed.)
    uint32_t bx, by, bxq, bxr, byq, byr;
    udivmod(lin, tilesY, by, bx); udivmod(bx, (uint32_t)nresY, bxq, bxr);
    const int ybase = (int)bxr + (int)bxq * RY * K;
    udivmod(by, (uint32_t)nres, byq, byr);
    const int lbase = (int)byr + (int)byq * CH * K;
    if (ybase >= N || lbase >= nzl) return;
    const int zbase = lbase + (int)f.z0;
    float py[RY], pz[CH];
#pragma unroll
    for (int j = 0; j < RY; ++j) py[j] = axis_pos(f.oy, ybase + j * K, f.vs);
#pragma unroll
    for (int j = 0; j < CH; ++j) pz[j] = axis_pos(f.oz, zbase + j * K, f.vs);
    for (uint32_t i = tid; i < (uint32_t)PXT; i += NT) PX[i] = i < (uint32_t)N ? axis_pos(f.ox, i, f.vs) : INFINITY;
    for (uint32_t i = tid; i < (uint32_t)TAB; i += NT) {
        if (i < (uint32_t)N) {
            const uint32_t si = scr(i);
            const float sy = axis_pos(f.oy, i, f.vs), sz = axis_pos(f.oz, i, f.vs);
#pragma unroll
            for (int j = 0; j < RY; ++j) { const float d = sy - py[j]; TY[j / EY][si][j % EY] = d * d; }
            if (WIDE) TZ[0][si][0] = sz;
            else {
#pragma unroll
                for (int j = 0; j < CHT; ++j) { const float d = sz - pz[j]; TZ[j / EZ][si][j % EZ] = d * d; }
            }
        } else {                                                   // slots no real id refers to ("none" does: TAB - 1); finite: inf + it = inf
#pragma unroll
            for (int j = 0; j < RY; ++j) TY[j / EY][i][j % EY] = 0.0f;
#pragma unroll
            for (int j = 0; j < CHT; ++j) TZ[j / EZ][i][j % EZ] = 0.0f;
        }
    }
    if constexpr (TABZ > TAB)
        for (uint32_t i = (uint32_t)TAB + tid; i < (uint32_t)TABZ; i += NT) TZ[0][i][0] = INFINITY;
    if (WIDE) {
        // row numbers of the tile's source rows: entry pj * NR + rr = plane zbase + (pj - 1) k, row ybase + (rr - 1) k (0 where outside)
        for (uint32_t i = tid; i < (uint32_t)((CH + 2) * NR); i += NT) {
            const int zg = zbase + ((int)(i / NR) - 1) * K, yy = ybase + ((int)(i % NR) - 1) * K;
            const bool ok = zg >= (int)f.z0 - K && zg < (int)f.z1 + K && zg >= 0 && zg < N && yy >= 0 && yy < N;
            RB[i] = ok ? (uint32_t)((zg - (int)f.z0) * N + yy) : 0u;   // negative for the minus halo planes of a slab: read back as int
        }
    }
    if (FINAL && !GM) {
        for (uint32_t i = tid; i < (uint32_t)(RY * CH) * f.w; i += NT) {
            uint32_t ou, c;
            udivmod(i, f.w, ou, c);                                // (the divisor is uniform; word indices stay below 2^28: 32-bit arithmetic)
            const int o = (int)ou, a = o / CH, j = o % CH;
            const int oy = ybase + a * K, oz = zbase + j * K;
            WM[o * (TAB / 32) + (int)c] = (oy < N && oz < (int)f.z1) ? words[(uint32_t)((oz - (int)f.z0) * N + oy) * f.w + c] : 0u;
        }
    }
    __syncthreads();

    const char* tx = reinterpret_cast<const char*>(PX);
    const char* ty = reinterpret_cast<const char*>(TY);
    const char* tz = reinterpret_cast<const char*>(TZ);
    const uint32_t rowBytes = (uint32_t)N * IDB;
    const size_t planeBytes = (size_t)N * rowBytes;
    // compact ids: byte planes of the source and the output volume and the byte row of "none".  `in` / `out` point at plane z0 of WHOLE
    // volumes (n^3 words, then n^3 bytes): byte plane z0 sits (n - z0) word planes + z0 byte planes further on.
    const size_t toBytes = (size_t)(N - (int)f.z0) * planeBytes + (size_t)f.z0 * ((size_t)N * N);
    const char* inB = reinterpret_cast<const char*>(in) + toBytes;
    char* outB = reinterpret_cast<char*>(out) + toBytes;
    const char* noneB = reinterpret_cast<const char*>(none_row) + (size_t)TAB * 4u;
    int yout = 1, nout = 1;
    if constexpr (FULL) { yout = RY; nout = CH; }
    else {
#pragma unroll
        for (int j = 1; j < RY; ++j) yout += (ybase + j * K < N) ? 1 : 0;
#pragma unroll
        for (int j = 1; j < CH; ++j) nout += (zbase + j * K < (int)f.z1) ? 1 : 0;
    }
    const uint32_t kb = k * IDB;
    // 32-bit ids: ranks and the gather are relative to the first source plane of the tile that lies in the grid (zlo below): at
    // most the whole volume, 4 GiB at n = 1024, so byte offset + 1 <= 2^32 - 3 fits the low word.  8-byte ids: see the header.

    const int zbase0 = zbase, lbase0 = lbase, ybase0 = ybase;
    const uint32_t xiters = ((uint32_t)N + NT - 1) / NT, xper = (xiters + xparts - 1) / xparts * NT;
    const uint32_t xbeg = xpart * xper, xend = min((uint32_t)N, xbeg + xper);
    for (uint32_t xb = xbeg; xb < xend; xb += NT) {
        uint32_t x = xb + tid;
        bool upper = false;                                        // PM: this lane is the x + k end of its pair
        if constexpr (PM != 0) {
            // pair index -> x: the pairs of a chain block of 2k voxels are (x0, x0 + k), x0 = block * 2k + (pair % k)
            uint32_t pl;
            if (PM == 1) { upper = tid & 1u; pl = tid >> 1; }
            else if (PM == 2) { upper = (tid >> 1) & 1u; pl = ((tid >> 2) << 1) | (tid & 1u); }
            else if (PM == 4) { const uint32_t l8 = tid & 7u; upper = l8 >> 2; pl = ((tid >> 3) << 2) | (upper ? 7u - l8 : l8); }
            else { upper = (tid >> 3) & 1u; pl = ((tid >> 4) << 3) | (tid & 7u); }
            const uint32_t pr = (xb >> 1) + pl;
            x = (((pr & ~(k - 1u)) << 1) | (pr & (k - 1u))) + (upper ? k : 0u);
        } else if (x >= xend) break;
        // The uniform bases are re-read through an empty asm in every x iteration: otherwise every address of the ~10
        // planes of the tile is hoisted out of the x loop, does not fit the SGPR file and is spilled to VGPR lanes
        // (v_readlane / v_writelane were 6 % of the VALU instructions of the loop).
        const int zbase = (int)opaque_uniform((size_t)(uint32_t)zbase0), lbase = (int)opaque_uniform((size_t)(uint32_t)lbase0);
        const int ybase = (int)opaque_uniform((size_t)(uint32_t)ybase0);
        const int zlo = max(zbase - K, 0);
        const char* gbase = reinterpret_cast<const char*>(in) + ((ptrdiff_t)zlo - (ptrdiff_t)f.z0) * (ptrdiff_t)planeBytes;
        uint32_t ro[NR];
        bool yv[NR];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int ny = ybase + (rr - HY) * K;
            yv[rr] = ny >= 0 && ny < N && max(rr - HY - 1, 0) < yout;
            ro[rr] = (uint32_t)(yv[rr] ? ny : 0) * rowBytes;
        }
        const float px = PX[x];
        const bool hasM = x >= k, hasP = x + k < (uint32_t)N;
        const uint32_t xo = x * IDB, xmo = hasM ? xo - kb : xo, xpo = hasP ? xo + kb : xo;   // a column outside the grid reads the centre column
        // PM: the partner's column (always inside the grid) and the column beyond the own one
        const uint32_t xpart = upper ? xo - kb : xo + kb, xout = upper ? xpo : xmo;
        const uint32_t xoB = xo >> 2, xmoB = xmo >> 2, xpoB = xpo >> 2, xoutB = xout >> 2;      // compact ids: the same columns in the byte plane
        // WIDE: rank of a candidate = (source row index << 14 | byte offset of the column it was read from) + 1

        // SKIP (wide passes, k >= n/4: half of the neighbour rows / planes / columns lie outside the grid): a source row that
        // does not exist is neither loaded nor evaluated (wave-uniform branch), and neither is a column x-k / x+k that no lane
        // of the wave has (it would re-evaluate the centre column, which cannot change the result).  Dense passes keep the
        // branch-free form: out-of-grid rows read a row of "none".
        const bool anyM = !SKIP || __any(hasM), anyP = !SKIP || __any(hasP);
        // A source plane: its base address (wave-uniform, computed once per plane, right where the plane is first used) and
        // whether it exists at all (inside the grid, and needed by an output plane of this tile that exists).
        struct Plane { const char* base; const char* baseB; bool ok; };
        auto plane_of = [&](int zg, bool needed) {
            Plane pl;
            pl.ok = needed && zg >= 0 && zg < N;
            pl.base = opaque_uniform(reinterpret_cast<const char*>(in) + ((ptrdiff_t)(pl.ok ? zg : (int)f.z0) - (ptrdiff_t)f.z0) * (ptrdiff_t)planeBytes);
            pl.baseB = CPT ? opaque_uniform(inB + ((ptrdiff_t)(pl.ok ? zg : (int)f.z0) - (ptrdiff_t)f.z0) * (ptrdiff_t)((size_t)N * N)) : nullptr;
            return pl;
        };
        // ids of row rr of a source plane -> w[rr*3 ..]: columns {x-k, x, x+k}; "none" where outside the grid or not needed
        auto load_row = [&](const Plane& pl, int rr, T (&w)[NI]) {
            const bool ok = pl.ok && yv[rr];
            if (SKIP && !ok) return;
#if defined(VP_ABL_HOT)                                                    // every source row = row 0 of the volume: loads hit the L1 (timing only)
            const __amdgpu_buffer_rsrc_t b = row_resource(ok ? reinterpret_cast<const char*>(in) : reinterpret_cast<const char*>(none_row), rowBytes);
#else
            const __amdgpu_buffer_rsrc_t b =
                row_resource(ok ? pl.base + ro[rr] : reinterpret_cast<const char*>(none_row), rowBytes);
#endif
            if constexpr (CPT) {
                // word and byte of the three columns (a row of the byte plane is a quarter of the word row)
                const __amdgpu_buffer_rsrc_t bb = row_resource(ok ? pl.baseB + (ro[rr] >> 2) : noneB, (uint32_t)N);
                if constexpr (PM != 0) {
                    uint32_t w0, w1;
                    row_load(w0, b, xo); row_load(w1, b, xout);
                    w[rr * 2 + 0] = make_uint2(w0, row_load_u8(bb, xoB));
                    w[rr * 2 + 1] = make_uint2(w1, row_load_u8(bb, xoutB));
                } else {
                    uint32_t w0, w1, w2;
                    row_load(w0, b, xmo); row_load(w1, b, xo); row_load(w2, b, xpo);
                    w[rr * 3 + 0] = make_uint2(w0, row_load_u8(bb, xmoB));
                    w[rr * 3 + 1] = make_uint2(w1, row_load_u8(bb, xoB));
                    w[rr * 3 + 2] = make_uint2(w2, row_load_u8(bb, xpoB));
                }
            } else if constexpr (PM != 0) {
                row_load(w[rr * 2 + 0], b, xo);
                row_load(w[rr * 2 + 1], b, xout);
            } else {
                if (anyM) row_load(w[rr * 3 + 0], b, xmo);
                row_load(w[rr * 3 + 1], b, xo);
                if (anyP) row_load(w[rr * 3 + 2], b, xpo);
            }
        };

        B best[RY][CH];

        // What an id turns into before its candidate steps: seed x, squared y differences to the output rows it serves, squared
        // z differences to the output planes -- one LDS lookup each.
        struct Dec { float sx; float dy2[RY]; float dz2[CH]; uint32_t zt; };     // zt (compact ids): top z bit << 1, part of the rank
        auto lookup = [&](int P, int rr, T id, Dec& d) {
            const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1), olo = max(P - 1, 0), ohi = min(P + 1, CH - 1);
            d.sx = lds_f32(tx + ID::xoff(id));                              // 32-bit ids: "none" reads slot TAB = +inf
            if constexpr (WIDE && !CPT) d.sx = ID::is_none(id) ? INFINITY : d.sx;   // (inf - px)^2 = inf: "none" loses every '<'
            if constexpr (CPT) d.zt = ID::zt2(id); else d.zt = 0u;
            const uint32_t yo = ID::yoff(id), zo = ID::zoff(id);
#if defined(VP_ABL_NOLDS)                                                  // VP_ABL_*: ablation builds for the pipe analysis of DESIGN.md section 4 (WRONG results, timing only)
            for (int a = 0; a < RY; ++a) d.dy2[a] = __uint_as_float(yo + a);
            for (int o = 0; o < CH; ++o) d.dz2[o] = __uint_as_float(zo + o);
#else
            lds_span<RY, EY, TAB>(ty, yo, alo, ahi, d.dy2);
            if constexpr (WIDE) {
                const float sz = lds_f32(tz + zo);                          // seed z position; the squares per output plane are formed here
#pragma unroll
                for (int o = olo; o <= ohi; ++o) { const float dzv = sz - pz[o]; d.dz2[o] = dzv * dzv; }
            } else {
                lds_span<CH, EZ, TAB>(tz, zo, olo, ohi, d.dz2);
            }
#endif
        };
#ifndef VP_FINAL_MIN3
#define VP_FINAL_MIN3 1
#endif
        float hold[3][3];                                                   // FINAL: distances of column x - k of the current source row
        // c = position of the candidate column in the sequence of a source row (0, 1, 2: the FINAL form pairs the first two in one
        // v_min3_f32); ownCol = it is the lane's own column (rank 0 for the own voxel); coloff = byte offset of the column in its row
        // (the rank); DPPV = the decoded values are the PARTNER's (pair mode): they are read through the DPP permutation
        auto steps_col = [&](int P, int rr, int c, bool ownCol, uint32_t coloff, const Dec& d, uint32_t prank, auto dppv) {
            constexpr bool DPPV = decltype(dppv)::value;
            auto val = [&](float v) { if constexpr (DPPV) return from_partner<PM ? PM : 1>(v); else return v; };
            auto valu = [&](uint32_t v) { if constexpr (DPPV) return __float_as_uint(from_partner<PM ? PM : 1>(__uint_as_float(v))); else return v; };
            const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1), olo = max(P - 1, 0), ohi = min(P + 1, CH - 1);
            const float dxv = val(d.sx) - px;
            const float dx2 = dxv * dxv;
            u32x2 cand;
            if (!FINAL) {
                if constexpr (CPT) cand.x = (uint32_t)(((P + 1) * NR + rr + 1) << 14) + 1u + coloff + valu(d.zt);
                else cand.x = (WIDE ? (uint32_t)(((P + 1) * NR + rr) << 14) + 1u : prank + ro[rr]) + coloff;
            }
#pragma unroll
            for (int a = alo; a <= ahi; ++a) {
                const float pre = val(d.dy2[a]) + dx2;
                const bool ownRow = (rr == a + HY) && ownCol;
#pragma unroll
                for (int o = olo; o <= ohi; ++o) {
                    const float dd = val(d.dz2[o]) + pre;
                    if constexpr (FINAL) {
                        // distances only, so the order of the candidates no longer matters: the left column's distance waits for
                        // the centre column's and both go through one v_min3_f32 (27 -> 18 minimum instructions per voxel)
                        if (VP_FINAL_MIN3 && c == 0) hold[a - alo][o - olo] = dd;
                        else if (VP_FINAL_MIN3 && c == 1) best[a][o] = min3_f32(best[a][o], hold[a - alo][o - olo], dd);
                        else best[a][o] = min_f32(best[a][o], dd);
                    } else {
                        u32x2 cd = cand;
                        if (ownRow && o == P) cd.x = CPT ? d.zt : 0u;      // the voxel's own state wins every tie (sequential.cpp:84,106)
                        cd.y = __float_as_uint(dd);
#if defined(VP_ABL_NOMIN)
                        { u32x2 t = __builtin_bit_cast(u32x2, best[a][o]); t.y = __float_as_uint(min_f32(__uint_as_float(t.y), dd)); best[a][o] = __builtin_bit_cast(double, t); }
#else
                        best[a][o] = min_f64(best[a][o], __builtin_bit_cast(double, cd));
#endif
                    }
                }
            }
        };
        // the candidate steps that follow the decode of id j of a plane (j = rr * NC + column slot)
        auto steps = [&](int P, int rr, int c, const Dec& d, uint32_t prank) {
            if constexpr (PM != 0) {
                if (c == 0) {                                                   // own column, then the partner's evaluation of ITS own column
                    steps_col(P, rr, 0, true, xo, d, prank, std::false_type{});
                    steps_col(P, rr, 1, false, xpart, d, prank, std::true_type{});
                } else {
                    steps_col(P, rr, 2, false, xout, d, prank, std::false_type{});
                }
            } else {
                steps_col(P, rr, c, c == 1, c == 0 ? xmo : c == 1 ? xo : xpo, d, prank, std::false_type{});
            }
        };
#ifndef VP_DENSE_PIPE
#define VP_DENSE_PIPE (ID::kTab == 512 || (FINAL && !WIDE))      // measured (tools/ab_step.py): -1.3 % at n = 512, +0.8 % at n = 1024 (dense); FINAL at n = 1024: see DEPTH
#endif
        auto scatter = [&](int P, T (&w)[NI]) {
            // rank of the ids of this plane: byte offset of the row inside the gather window + 1 (wave-uniform) + the column offset
            // (WIDE: (source row index << 11) + x + 1, the row part added per row in steps() through the unrolled constant below)
            const uint32_t prank = WIDE ? 0u : (uint32_t)((zbase + P * K - zlo) * (ptrdiff_t)planeBytes) + 1u;
            const int olo = max(P - 1, 0), ohi = min(P + 1, CH - 1);
            const bool curOk = P <= nout && zbase + P * K >= 0 && zbase + P * K < N;      // SKIP: does this source plane exist
            Plane next{nullptr, nullptr, false};
            if (ROLL && P + 1 <= CH - (CZ ? 1 : 0)) next = plane_of(zbase + (P + 1) * K, P + 1 <= nout);
            if (VP_DENSE_PIPE && !SKIP) {
                // Software pipeline over the 18 ids of the plane: the table lookups of id j + 1 are issued BEFORE the candidate
                // steps of id j (the scheduling barriers keep the compiler from sinking them back to their first use), so a wave
                // waits for LDS data a whole id of VALU work after asking for it instead of immediately.
#ifndef VP_DENSE_PIPE_DEPTH
                // ids looked up ahead of the one being evaluated.  Two ahead (profiles/r03/ab_pipe_*.txt): fused last pass at n = 1024
                // 3.23 -> 3.05 ms (-6 %), at n = 512 +-0; dense passes +1 % (n = 512) and +24 % (n = 1024: 7 more VGPRs cost a workgroup per CU)
#define VP_DENSE_PIPE_DEPTH ((FINAL && ID::kTab == 1024) ? 2 : 1)
#endif
                constexpr int DEPTH = VP_DENSE_PIPE_DEPTH;
                Dec d[DEPTH + 1];
#pragma unroll
                for (int j = 0; j < DEPTH && j < NI; ++j) lookup(P, j / NC, w[j], d[j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int rr = j / NC, c = j % NC;
                    if (j + DEPTH < NI) lookup(P, (j + DEPTH) / NC, w[j + DEPTH], d[(j + DEPTH) % (DEPTH + 1)]);
                    __builtin_amdgcn_sched_barrier(0);
                    steps(P, rr, c, d[j % (DEPTH + 1)], prank);
                    if (c == NC - 1) {
                        const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1);
#pragma unroll
                        for (int a = alo; a <= ahi; ++a)
#pragma unroll
                            for (int o = olo; o <= ohi; ++o) pin(best[a][o]);
                        if (ROLL && P + 1 <= CH - (CZ ? 1 : 0)) load_row(next, rr, w);     // rolling prefetch (see below)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1);
                if (!SKIP || (curOk && yv[rr])) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (SKIP && ((c == 0 && !anyM) || (c == 2 && !anyP))) continue;
                        Dec d;
                        lookup(P, rr, w[rr * NC + c], d);
                        steps(P, rr, c, d, prank);
                    }
#pragma unroll
                    for (int a = alo; a <= ahi; ++a)
#pragma unroll
                        for (int o = olo; o <= ohi; ++o) pin(best[a][o]);
                }
                // Rolling prefetch: the three ids of this row are spent, so the same row of the NEXT source plane is
                // requested into their registers right away -- a whole plane of evaluation ahead of its use, without a
                // second id buffer (18 VGPRs).  A plane that is not needed reads "none" (never memory past the slab's halo).
                if (ROLL && P + 1 <= CH - (CZ ? 1 : 0)) load_row(next, rr, w);
            }
        };

        // ROLL: one id buffer, refilled row by row (see scatter).  Otherwise two buffers: plane P + 1 is requested as a whole
        // before plane P is evaluated (18 more VGPRs).  Measured (tools/ab_pass.py, interleaved): at n = 512 the two
        // buffers are 3 % ahead, at n = 1024 the rolling refill is 5 % ahead.
        T w[NI], w2[ROLL ? 1 : NI];
        T pend[RY];                                                // gathered winners of the previous output plane, stored a plane later
        {
            const Plane first = plane_of(CZ ? zbase : zbase - K, true);   // closed tiles start at their own first plane
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) load_row(first, rr, w);
        }
#pragma clang loop unroll(full)
        for (int P = -1; P <= CH; ++P) {
            T (&cur)[NI] = (ROLL || !((P + 1) & 1)) ? w : reinterpret_cast<T (&)[NI]>(w2);
            if (!ROLL && P + 1 <= CH - (CZ ? 1 : 0) && !(CZ && P == -1)) {
                T (&nxt)[NI] = ((P + 1) & 1) ? w : reinterpret_cast<T (&)[NI]>(w2);
                const Plane np = plane_of(zbase + (P + 1) * K, P + 1 <= nout);
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) load_row(np, rr, nxt);
            }
            if (P + 1 < CH) {
#pragma unroll
                for (int a = 0; a < RY; ++a) {
                    if constexpr (FINAL) best[a][P + 1] = INFINITY;
                    else best[a][P + 1] = __builtin_bit_cast(double, (u32x2){0xFFFFFFFFu, 0x7F800000u});   // (+inf, last rank)
                }
            }
            uint32_t mw[RY] = {};
            if constexpr (FINAL && GM) {         // bitmask words of the rows stored after this plane: requested before its evaluation
                if (P >= 1 && P - 1 < nout) {
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        const size_t rowIdx = (size_t)(opaque_uniform((size_t)lbase) + (P - 1) * K) * N + (ybase + a * K);
                        mw[a] = words[rowIdx * f.w + (x >> 5)];
                    }
                }
            }
            // The planes before the first and after the last output plane may lie outside the grid (one of them does for every tile at
            // k = n/16, for half the tiles at n/32, ...): then they hold nothing but "none" and could be skipped with a workgroup-uniform
            // branch.  Measured: a branch around EVERY plane and around the first / last row +1.7 % (and spills with the rows), a branch at
            // the two ends of the chain only -0.3 % / +0.3 % (profiles/r03/ab_edge_*.txt, ab_endskip_*.txt) -- removed: what pays is
            // dropping the halo at compile time, which only the closed tiles of k = n/8 can.
            if (!(CZ && (P == -1 || P == CH))) scatter(P, cur);      // closed tiles have no plane before the first or after the last
            if constexpr (FINAL) {
                if (P >= 1 && P - 1 < nout) {
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        const size_t rowIdx = (size_t)(opaque_uniform((size_t)lbase) + (P - 1) * K) * N + (ybase + a * K);
                        const bool set = ((GM ? mw[a] : WM[(a * CH + (P - 1)) * (TAB / 32) + (x >> 5)]) >> (x & 31)) & 1u;
                        row_store<VP_DENSE_STORE_AUX>(set ? best[a][P - 1] : copysignf(best[a][P - 1], fill), row_resource(sdf + rowIdx * N, (uint32_t)N * 4u), x * 4u);
                    }
                }
            } else {
                if (P >= 2 && P - 2 < nout) {                      // ids gathered during the previous plane
                    const char* orow = opaque_uniform(reinterpret_cast<const char*>(out) + ((size_t)(lbase + (P - 2) * K) * N + ybase) * rowBytes);
                    const char* orowB = CPT ? opaque_uniform(outB + ((size_t)(lbase + (P - 2) * K) * N + ybase) * (size_t)N) : nullptr;
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        if constexpr (CPT) {
                            row_store<VP_DENSE_STORE_AUX>(pend[a].x, row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                            row_store_u8<VP_DENSE_STORE_AUX>(pend[a].y, row_resource(orowB + (size_t)(a * K) * N, (uint32_t)N), xoB);
                        } else {
                            row_store<VP_DENSE_STORE_AUX>(pend[a], row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                        }
                    }
                }
                if (P >= 1 && P - 1 < nout) {                      // output plane P - 1 is complete: fetch the ids of its winners
                    const uint32_t orank = (uint32_t)((zbase + (P - 1) * K - zlo) * (ptrdiff_t)planeBytes);
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        const uint32_t lo = __builtin_bit_cast(u32x2, best[a][P - 1]).x;
                        if constexpr (CPT) {
                            // rank = (source row index + 1) << 14 | byte offset in the word row, + 1, + top z bit << 1; the own voxel (rank < 4) sits
                            // in row (P, a + 1) of the tile's source rows.  The byte of the output: top z bit from the rank, "none" iff nothing won.
                            const uint32_t zt = lo & 2u;
                            const uint32_t r = lo < 4u ? (uint32_t)((P * NR + a + 1 + 1) << 14) + xo : lo - 1u - zt;
                            const ptrdiff_t row = (ptrdiff_t)(int)RB[(r >> 14) - 1u];
                            const uint32_t none1 = __builtin_bit_cast(u32x2, best[a][P - 1]).y == 0x7F800000u ? IdC::kNoneBit : 0u;
                            pend[a] = make_uint2(*reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(in) + row * (ptrdiff_t)rowBytes + (ptrdiff_t)(r & 16383u)),
                                                 zt | none1);
                        } else if constexpr (WIDE) {
                            // rank - 1 = source row index << 14 | byte offset in the row; the own voxel (rank 0) sits in row (P, a + 1) of the tile's source rows
                            const uint32_t r = lo ? lo - 1u : (uint32_t)((P * NR + a + 1) << 14) + xo;
                            const ptrdiff_t row = (ptrdiff_t)(int)RB[r >> 14];
#if defined(VP_ABL_NOGATHER)
                            pend[a] = ID::pack(r & 2047u, (uint32_t)row & 2047u, 0u);
#else
                            pend[a] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(in) + row * (ptrdiff_t)rowBytes + (ptrdiff_t)(r & 16383u));
#endif
                        } else {
                            const uint32_t ownOff = orank + ro[a + HY] + xo;
                            const uint32_t off = lo ? lo - 1u : ownOff;
#if defined(VP_ABL_NOGATHER)
                            pend[a] = T(off);
#elif defined(VP_ABL_HOT)
                            pend[a] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(in) + (off & (rowBytes - 1u) & ~3u));
#else
#if VP_DENSE_GATHER_NT
                            pend[a] = __builtin_nontemporal_load(reinterpret_cast<const T*>(gbase + off));
#else
                            pend[a] = *reinterpret_cast<const T*>(gbase + off);
#endif
#endif
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < RY; ++a)
#pragma unroll
                for (int o = P; o <= P + 1; ++o)
                    if (o >= 0 && o < CH) pin(best[a][o]);
        }
        if constexpr (!FINAL) {
            if (CH - 1 < nout) {
                const char* orow = opaque_uniform(reinterpret_cast<const char*>(out) + ((size_t)(lbase + (CH - 1) * K) * N + ybase) * rowBytes);
                const char* orowB = CPT ? opaque_uniform(outB + ((size_t)(lbase + (CH - 1) * K) * N + ybase) * (size_t)N) : nullptr;
#pragma unroll
                for (int a = 0; a < RY; ++a) {
                    if (a >= yout) continue;
                    if constexpr (CPT) {
                        row_store<VP_DENSE_STORE_AUX>(pend[a].x, row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                        row_store_u8<VP_DENSE_STORE_AUX>(pend[a].y, row_resource(orowB + (size_t)(a * K) * N, (uint32_t)N), xoB);
                    } else {
                        row_store<VP_DENSE_STORE_AUX>(pend[a], row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ seed scatter: one proposal round
// The seed at (sx, sy, sz), held by the voxel of slot s of a closed 4-chain tile (slot = ((plane * 4 + row) * 4 + segment) * XR +
// residue), proposes itself to the voxels STEP chain positions away along each axis, s itself included (rank 0).
//     key = distance bits << 32 | rank << 27 | tag,   rank = 1 + scan index of s as seen from the target (sequential.cpp:84-112)
// (tag: what the winner is to be known by -- the slot s, or the slot of the seed it carries; it never decides a comparison, since two
// proposals to one target with equal rank come from the same slot)
// The minimum is commutative, so the loops run (row, column, plane): dx^2 + dy^2 is formed once per row and column.
// The minimum is commutative, so the loops run (row, column, plane): dx^2 + dy^2 is formed once per row and column.  (Spreading
// the 27 proposals of a seed over nine threads was measured: -3 % at n = 1024, +5 .. 14 % at n = 512, profiles/r02/ab34.txt.)
// An empty slot holds (+inf, rank 0, slot 0): every key with a finite distance is smaller, and a proposal whose distance is not
// finite (sequential.cpp:106 never takes such a candidate) is >= it and leaves the slot as it is -- no test needed.
constexpr unsigned long long kEmptyKey = 0x7F80000000000000ull;

// STEP = 2 (the pass with k = n/2 inside a closed 4-chain tile): along each axis a voxel has exactly two in-grid chain positions two
// steps apart -- its own and position ^ 2 -- so a seed has 8 targets, all valid: no tests, no branches (the general form below
// compiles to 27 predicated blocks of which a wave of border voxels executes every one).  Scan index of s as seen from the partner
// along an axis: partner = position - t * 2 with t = +1 if position >= 2 else -1, i.e. t + 1 = position & 2.
// Measured (profiles/r03/ab_propose_*.txt): jfa_first_two 0.339 -> 0.311 ms at n = 512, 2.56 -> 2.37 ms at n = 1024.  The same idea for
// STEP = 1 -- 27 straight-line minima, targets outside the chain neutralised with the key ~0 -- trades 4 SALU for 2 VALU per candidate
// and gave the gain back (0.338 / 2.53): the kernel is bound by vector issue of the one wave that proposes.
// Where the positions of a tile's chain members come from: computed (cvt, mul, add and the index arithmetic before them: ~5
// instructions per position, nine positions per proposal round), or read from three small LDS tables the tile fills once
// (jfa_first_two: the proposing wave's instruction stream is the tile's critical path).  Chain positions outside 0..3 are only ever
// asked for targets that are then skipped; the tables wrap them, the arithmetic lets them run wild.
struct ChainPosCalc {
    const Frame& f; uint32_t rx0, ry, rz, k;
    __device__ __forceinline__ float x(uint32_t seg, uint32_t xr) const { return axis_pos(f.ox, rx0 + xr + __umul24(seg, k), f.vs); }
    __device__ __forceinline__ float y(uint32_t j) const { return axis_pos(f.oy, ry + __umul24(j, k), f.vs); }
    __device__ __forceinline__ float z(uint32_t j) const { return axis_pos(f.oz, rz + __umul24(j, k), f.vs); }
};
template <int XR>
struct ChainPosLds {
    const float* px; const float* py; const float* pz;             // [4 * XR], [4], [4]
    __device__ __forceinline__ float x(uint32_t seg, uint32_t xr) const { return px[(seg & 3u) * XR + xr]; }
    __device__ __forceinline__ float y(uint32_t j) const { return py[j & 3u]; }
    __device__ __forceinline__ float z(uint32_t j) const { return pz[j & 3u]; }
};

template <int XR, class POS>
__device__ __forceinline__ void propose_half(unsigned long long* keys, uint32_t s, float sx, float sy, float sz, const POS& pos)
{
    const uint32_t xr = s % XR, xs = (s / XR) & 3u, jr = (s / (4u * XR)) & 3u, jp = s / (16u * XR);
    float dx2[2], dy2[2], dz2[2];
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {                             // 0: the seed's own chain position, 1: the partner's
        const float dxv = sx - pos.x(xs ^ (2u * j), xr);
        const float dyv = sy - pos.y(jr ^ (2u * j));
        const float dzv = sz - pos.z(jp ^ (2u * j));
        dx2[j] = dxv * dxv; dy2[j] = dyv * dyv; dz2[j] = dzv * dzv;
    }
    const uint32_t ra[2] = {1u, xs & 2u}, rb[2] = {3u, (jr & 2u) * 3u}, rc[2] = {9u, (jp & 2u) * 9u};      // (t + 1) * {1, 3, 9}
#pragma unroll
    for (uint32_t ib = 0; ib < 2; ++ib)
#pragma unroll
        for (uint32_t ia = 0; ia < 2; ++ia) {
            const float pre = dx2[ia] + dy2[ib];
#pragma unroll
            for (uint32_t ic = 0; ic < 2; ++ic) {
                const float d = pre + dz2[ic];
                const uint32_t rank = (ia | ib | ic) ? rc[ic] + rb[ib] + ra[ia] + 1u : 0u;
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | ((rank << 27) | s);
                __hip_atomic_fetch_min(&keys[s ^ (ia * 2u * XR) ^ (ib * 8u * XR) ^ (ic * 32u * XR)], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
}

template <int XR, int STEP, class POS>
__device__ __forceinline__ void propose(unsigned long long* keys, uint32_t s, float sx, float sy, float sz, const POS& pos, uint32_t tag)
{
    const uint32_t xr = s % XR, xs = (s / XR) & 3u, jr = (s / (4u * XR)) & 3u, jp = s / (16u * XR);
    float dx2[3], dy2[3], dz2[3];
    bool va[3], vb[3], vc[3];
#pragma unroll
    for (int t = -1; t <= 1; ++t) {                                // target = s - t STEP positions: s is its neighbour at +t
        va[t + 1] = xs - t * STEP <= 3u; vb[t + 1] = jr - t * STEP <= 3u; vc[t + 1] = jp - t * STEP <= 3u;   // unsigned: also rejects < 0
        const float dxv = sx - pos.x(xs - t * STEP, xr);
        const float dyv = sy - pos.y(jr - t * STEP);
        const float dzv = sz - pos.z(jp - t * STEP);
        dx2[t + 1] = dxv * dxv; dy2[t + 1] = dyv * dyv; dz2[t + 1] = dzv * dzv;
    }
#pragma unroll
    for (int b = -1; b <= 1; ++b) {
        if (!vb[b + 1]) continue;
#pragma unroll
        for (int a = -1; a <= 1; ++a) {
            if (!va[a + 1]) continue;
            const float pre = dx2[a + 1] + dy2[b + 1];
            // lowest plane first: the three planes are then constant non-negative offsets from one index
            const uint32_t t0 = s - (uint32_t)(b * STEP) * (4u * XR) - (uint32_t)(a * STEP) * XR - (uint32_t)STEP * (16u * XR);
#pragma unroll
            for (int c = -1; c <= 1; ++c) {
                if (!vc[c + 1]) continue;
                const float d = pre + dz2[c + 1];
                const bool own = a == 0 && b == 0 && c == 0;       // (a distance that is not finite never wins: kEmptyKey)
                const uint32_t rank = own ? 0u : (uint32_t)((c + 1) * 9 + (b + 1) * 3 + (a + 1) + 1);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | ((rank << 27) | tag);
                __hip_atomic_fetch_min(&keys[t0 + (uint32_t)((1 - c) * STEP) * (16u * XR)], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ seed scatter (k = n/4)
// The pass with k = n/4 sees a state in which few voxels hold a seed yet (5 % on the headline mesh; tools/seed_occupancy.py),
// but three quarters of the 64-voxel row segments hold at least one, so wave-uniform skipping barely helps the gather form:
// it evaluates 27 candidate slots per voxel and 87 % of a dense pass's instructions.  This kernel turns the pass around.
// With k = n/4 the chain {r, r+k, r+2k, r+3k} along an axis is closed under +-k, so a tile of 4 planes x 4 rows x 4 x-segments
// (XR residues wide) reads exactly the voxels it writes.  The tile's ids go to LDS, the few that are seeds are compacted into a
// list, and every list entry PROPOSES itself to the up to 27 voxels it is a candidate of with one 64-bit LDS minimum
//     key = distance bits << 32 | rank << 27 | source slot,   rank 0 = the voxel's own state, 1 + scan index otherwise,
// which is the reference's "first minimum in scan order, own state first" (sequential.cpp:84-112) for the same reason as in
// jfa_pass_dense.  Work is proportional to the seeds, not to the voxels; traffic is one read and one write of the id volume.
template <class ID, int XR, int NT>
__global__ void __launch_bounds__(NT)
jfa_pass_seeds(Frame f, uint32_t k, const typename ID::T* __restrict__ in, typename ID::T* __restrict__ out)
{
    using T = typename ID::T;
    constexpr uint32_t SLOTS = 64u * XR;                           // slot = ((plane * 4 + row) * 4 + segment) * XR + residue
    constexpr unsigned long long kEmpty = kEmptyKey;
    __shared__ unsigned long long keys[SLOTS];
    __shared__ T ids[SLOTS];
    __shared__ uint16_t list[SLOTS];
    __shared__ uint32_t cnt;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, N = f.n;
    const uint32_t rx0 = blockIdx.x * XR, ry = blockIdx.y, rz = blockIdx.z;
    if (tid == 0) cnt = 0;
    __syncthreads();
    constexpr int PER = (int)(SLOTS / NT);                         // slots per thread, all requested before any is used
    // slot tid + i NT = row-plane (rpb + i G) x column `col`: a thread keeps its x, and the row-plane of an iteration is the same
    // for the whole wave (4 XR >= 64 slots per row-plane), so row addresses are scalar work
    constexpr uint32_t RPW = 4u * XR;
    constexpr uint32_t G = NT / RPW;
    static_assert(NT % RPW == 0 && RPW % 64u == 0, "a wave must stay inside one row-plane");
    const uint32_t col = tid % RPW, myx = rx0 + col % XR + __umul24(col / XR, k);
    const bool xin = rx0 + col % XR < k;
    const uint32_t rpb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / RPW));
    auto row_of = [&](int i) {                                     // voxel index of x = 0 of the row-plane of iteration i
        const uint32_t rp = rpb + (uint32_t)i * G;
        return ((size_t)(rz + (rp >> 2) * k) * N + (ry + (rp & 3u) * k)) * N;
    };
    T mine[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        mine[i] = ID::none();
        if (xin) mine[i] = in[row_of(i) + myx];
    }
    uint32_t nmine = 0;                                            // seeds of this wave, lane 0 reserves list space once
    unsigned long long ms[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t s = tid + (uint32_t)i * NT;
        keys[s] = kEmpty;
        ids[s] = mine[i];
        ms[i] = __builtin_amdgcn_ballot_w64(!ID::is_none(mine[i]));
        nmine += (uint32_t)__popcll(ms[i]);
    }
    uint32_t base = 0;
    if (lane == 0 && nmine) base = atomicAdd(&cnt, nmine);
    base = (uint32_t)__shfl((int)base, 0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (!ID::is_none(mine[i])) list[base + (uint32_t)__popcll(ms[i] & ((1ull << lane) - 1ull))] = (uint16_t)(tid + (uint32_t)i * NT);
        base += (uint32_t)__popcll(ms[i]);
    }
    __syncthreads();
    const uint32_t nseeds = cnt;
    for (uint32_t e = tid; e < nseeds; e += NT) {
        const uint32_t s = list[e];
        const T id = ids[s];
        propose<XR, 1>(keys, s, axis_pos(f.ox, ID::xoff(id) >> 2, f.vs), axis_pos(f.oy, scr(ID::yoff(id) >> 2), f.vs),
                       axis_pos(f.oz, scr(ID::zoff(id) >> 2), f.vs), ChainPosCalc{f, rx0, ry, rz, k}, s);
    }
    __syncthreads();
    unsigned long long won[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) won[i] = keys[tid + (uint32_t)i * NT];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (!xin) continue;
        const T id = won[i] == kEmpty ? ID::none() : ids[(uint32_t)won[i] & 0x07FFFFFFu];
        out[row_of(i) + myx] = id;
    }
}

// ------------------------------------------------------------------------------------------ first two passes from the mask
// Passes k = n/2 and k = n/4 in one kernel, straight from the border bitmask.  Both steps are steps along the closed 4-chains of
// jfa_pass_seeds (+-n/2 = two chain positions, +-n/4 = one), so the state after the first pass of a tile's 64 x XR voxels depends
// on the border bits of those same voxels only: nothing but 64 x XR bits is read, the first pass never touches HBM at all, and the
// id volume is written once.  Stage A scatters the border voxels (0.7 % on the headline mesh) two positions along each axis, which
// leaves every voxel with the slot of its pass-1 seed (or none); stage B scatters the voxels that have one (5 %) one position
// along each axis with that seed's coordinates.  Keys and ranks as in jfa_pass_seeds; a seed's coordinates are those of its slot.
// dev (-DVP_FIRST_TWO_TIMING, tools/first_two_stages.py): s_memtime stamps of thread 0 at the stage boundaries of every tile
#ifdef VP_FIRST_TWO_TIMING
constexpr unsigned kFtSlots = 1u << 17;
__device__ unsigned long long g_ft_slots[kFtSlots][16];          // one row per workgroup (modulo): plain stores, no atomics
#define VP_FT_STAMP(i) unsigned long long ft_t##i = 0; if (threadIdx.x == 0) ft_t##i = __builtin_amdgcn_s_memtime()
#define VP_FT_FLUSH(tile_) do { if (threadIdx.x == 0) { const unsigned long long t_[13] = {ft_t0, ft_t1, ft_t2, ft_t3, ft_t4, ft_t5, ft_t6, ft_t7, ft_t8, ft_t9, ft_t10, ft_t11, ft_t12}; \
        const unsigned w_ = (tile_) % kFtSlots; \
        for (int i_ = 1; i_ < 13; ++i_) g_ft_slots[w_][i_] = t_[i_] - t_[i_ - 1]; g_ft_slots[w_][0] = 1ull; } } while (0)
#else
#define VP_FT_STAMP(i) do {} while (0)
#define VP_FT_FLUSH(tile_) do {} while (0)
#endif
// Measured and dropped (round 3, profiles/r03/ab_step_512.txt): a PERSISTENT form of this kernel -- 8 workgroups per CU walking the
// tile sequence, the next tile's mask words requested a tile ahead -- ran 0.493 ms against 0.404 (n = 512) and 3.44 against 2.92
// (n = 1024).  The launch already keeps 7.5 of 8 wave slots per SIMD occupied (SQ_WAVE_CYCLES is in quad-cycles), so there was no
// dispatch gap to close, and workgroups that start together walk their five stages in step and meet at the LDS.
#ifndef VP_PROPOSE_HALF
#define VP_PROPOSE_HALF 1
#endif
#ifndef VP_FT_POS_LDS
#define VP_FT_POS_LDS 1
#endif
#ifndef VP_FT_FAST
#define VP_FT_FAST 1              // tiles whose lattices hold at most one border voxel each are written straight from a census (see the kernel)
#endif
#ifndef VP_FT_STORE_NT
#define VP_FT_STORE_NT 1          // the id volume leaves with the nt policy (it is read again only by the next pass, after all of it has been written):
                                  // jfa_first_two 0.238 -> 0.222 ms at n = 512, 1.61 -> 1.50 ms at n = 1024 (two boxes); at n = 2048 16.3 -> 12.5 ms on
                                  // one box and 17.6 -> 17.9 on another (profiles/r04/ab_ftnt_*.txt)
#endif
template <class T> __device__ __forceinline__ void ft_store(T* p, T v) { if (VP_FT_STORE_NT) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void ft_store(uint2* p, uint2 v)
{
    if (VP_FT_STORE_NT) __builtin_nontemporal_store(__builtin_bit_cast(unsigned long long, v), reinterpret_cast<unsigned long long*>(p)); else *p = v;
}
// CPT (round 4): the result leaves in the compact layout of IdC (word plane + byte plane) instead of ID's own; inside the kernel the
// ids stay ID's (Id64).
template <class ID, int XR, int NT, int TPW, bool CPT = false>
__global__ void __launch_bounds__(NT)
jfa_first_two(Frame f, const uint32_t* __restrict__ border, typename ID::T* __restrict__ out, uint32_t tilesX, uint32_t tiles, uint32_t shifts,
              FastDiv divTilesX, FastDiv divK)
{
    static_assert(!CPT || std::is_same<ID, Id64>::value, "compact output: from 8-byte ids");
    using T = typename ID::T;
    constexpr uint32_t SLOTS = 64u * XR;                           // slot = ((plane * 4 + row) * 4 + segment) * XR + residue
    constexpr int PER = (int)(SLOTS / NT);
    constexpr unsigned long long kEmpty = kEmptyKey;
    static_assert(SLOTS <= 0x10000u && SLOTS % NT == 0, "tile");
    __shared__ unsigned long long keys[SLOTS];
    __shared__ T idOf[SLOTS];                                      // packed id of the voxel of a slot (what a seed at that slot is called)
    __shared__ uint32_t list[SLOTS];                               // entries: slot | slot of the seed it holds << 16
    __shared__ float posX[4 * XR], posY[4], posZ[4];
    __shared__ uint32_t cnt[2];
    __shared__ uint32_t latCnt[TPW][XR], latSeed[TPW][XR];         // VP_FT_FAST: border voxels per lattice of each tile, the slot of one of them
    const uint32_t tid = threadIdx.x, lane = tid & 63u, N = f.n, k = N / 4u;
    // slot tid + i NT = row-plane (rpb + i G) x column `col` (see jfa_pass_seeds): row addresses are scalar work
    constexpr uint32_t RPW = 4u * XR;
    constexpr uint32_t G = NT / RPW;
    static_assert(NT % RPW == 0 && RPW % 64u == 0, "a wave must stay inside one row-plane");
    const uint32_t col = tid % RPW;
    const uint32_t rpb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / RPW));
    // wave-cooperative append of the slots whose flag is set (one reservation per wave)
    auto append = [&](const bool (&flag)[PER], const uint32_t (&seed)[PER], uint32_t& counter) {
        unsigned long long ms[PER];
        uint32_t n = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { ms[i] = __builtin_amdgcn_ballot_w64(flag[i]); n += (uint32_t)__popcll(ms[i]); }
        if (n == 0) return;                                        // uniform; every lane of the wave is here (no divergence above)
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counter, n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(ms[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms[i], 0u));
            if (flag[i]) list[base + below] = (tid + (uint32_t)i * NT) | (seed[i] << 16);
            base += (uint32_t)__popcll(ms[i]);
        }
    };
    // shifts = log2(tilesX) | log2(k) << 8 | 1 << 16 when both are powers of two: two run-time divisions per tile are ~50 instructions
    // of a wave that executes ~450 in all (the kernel is issue-bound, see profiles/r03/first_two_stages_tpw2_n512.txt)
    auto tile_origin = [&](uint32_t t, uint32_t& rx0, uint32_t& ry, uint32_t& rz) {
        if (shifts >> 16) {
            rx0 = (t & (tilesX - 1u)) * XR; const uint32_t q = t >> (shifts & 31u); ry = q & (k - 1u); rz = q >> ((shifts >> 8) & 31u);
        } else {                                                   // sides that are not powers of two: multiply-shift division (see FastDiv)
            uint32_t q, r;
            divTilesX.divmod(t, q, r); rx0 = r * XR;
            divK.divmod(q, rz, ry);
        }
    };
    // A workgroup works through TPW consecutive tiles (default 1).  Their border words -- the only thing read from memory, and 37 % of
    // a one-tile workgroup's life spent waiting for them (profiles/r03/first_two_stages_n512.txt) -- are ALL requested before the
    // first tile is touched.  (Nothing is loaded inside the tile loop, so no wait in it ever covers the stores of the tile before:
    // that is what made the persistent form of round 3 slower.)  More tiles per workgroup bought nothing, see VP_FIRST_TWO_TPW.
    const uint32_t tile0 = (blockIdx.x) * TPW;
    uint32_t mw[TPW][PER];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        uint32_t rx0, ry, rz;
        tile_origin(min(tile0 + u, tiles - 1u), rx0, ry, rz);
        const uint32_t myx = rx0 + col % XR + __umul24(col / XR, k);
        const bool xin = rx0 + col % XR < k;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t rp = rpb + (uint32_t)i * G, y = ry + (rp & 3u) * k, z = rz + (rp >> 2) * k;
            mw[u][i] = xin ? border[(z * N + y) * f.w + (myx >> 5)] : 0u;          // < 2^28 words at n = 2048: 32-bit index arithmetic
        }
    }
    if (VP_FT_FAST) {                                              // lattice census of every tile of this workgroup (see below); the border words are in flight
        for (uint32_t i = tid; i < (uint32_t)(TPW * XR); i += NT) (&latCnt[0][0])[i] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const uint32_t tile = tile0 + u;
        if (tile >= tiles) break;                                  // uniform
        uint32_t rx0, ry, rz;
        tile_origin(tile, rx0, ry, rz);
        const uint32_t myx = rx0 + col % XR + __umul24(col / XR, k);
        const bool xin = rx0 + col % XR < k;
        VP_FT_STAMP(0);
#if VP_FT_FAST
        // The 64 voxels of one residue class (x, y, z mod k) -- a LATTICE, 4 x 4 x 4 chain positions -- only ever see each other in these
        // two passes.  A lattice without a border voxel stays "none"; a lattice with exactly ONE ends with that seed in all 64 voxels
        // whatever the distances are (after the step of two chain positions the seed sits at {s, s ^ 2} along every axis, and every
        // position 0 .. 3 has one of those within one step).  Only lattices with two or more border voxels need the proposals -- 7 % of
        // them at n = 512 on the benchmark mesh, and 61 % of the tiles have none (profiles/r04/first_two_lattices.txt): those tiles write
        // their result straight from the census below: one barrier, no lists, keys or proposals.  With that much less to issue the wait
        // for the border words shows again, and two tiles per workgroup (all loads up front) pay: 0.305 -> 0.269 -> 0.239 ms at n = 512,
        // 2.17 -> 1.77 -> 1.59 ms at n = 1024, 22.3 -> 17.2 ms at n = 2048 (profiles/r04/ab_ftfast_*.txt, ab_fttpw_*.txt).
        {
            const uint32_t res = col % XR;
            uint32_t mine = 0, mySlot = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (xin && ((mw[u][i] >> (myx & 31u)) & 1u)) { ++mine; mySlot = tid + (uint32_t)i * NT; }
            bool multi = false;
            if (mine) {
                multi = atomicAdd(&latCnt[u][res], mine) + mine >= 2u;   // whoever adds last to a lattice of two or more sees it
                latSeed[u][res] = mySlot;
            }
            if (!__syncthreads_or(multi ? 1 : 0)) {
                const uint32_t sSlot = latSeed[u][res];               // valid where latCnt[u][res] == 1
                const T one = ID::pack(rx0 + res + __umul24((sSlot / XR) & 3u, k), ry + ((sSlot / (4u * XR)) & 3u) * k, rz + (sSlot / (16u * XR)) * k);
                const T id = latCnt[u][res] ? one : ID::none();
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    if (!xin) continue;
                    const uint32_t rp = rpb + (uint32_t)i * G;
                    const size_t vox = (size_t)((rz + (rp >> 2) * k) * N + (ry + (rp & 3u) * k)) * N + myx;
                    if constexpr (CPT) {
                        const uint2 c = IdC::from64(id);
                        ft_store(reinterpret_cast<uint32_t*>(out) + vox, c.x);
                        ft_store(reinterpret_cast<unsigned char*>(reinterpret_cast<uint32_t*>(out) + (size_t)N * N * N) + vox, (unsigned char)c.y);
                    } else {
                        ft_store(out + vox, id);
                    }
                }
                continue;                                              // next tile of the workgroup (uniform)
            }
        }
#endif
        if (tid < 2) cnt[tid] = 0;
        if (VP_FT_POS_LDS) {                                       // positions of the tile's 4 XR columns, 4 rows, 4 planes (see ChainPosLds)
            if (tid < RPW) posX[tid] = axis_pos(f.ox, myx, f.vs);  // tid < RPW: col == tid
            if (tid < 4) { posY[tid] = axis_pos(f.oy, ry + tid * k, f.vs); posZ[tid] = axis_pos(f.oz, rz + tid * k, f.vs); }
        }
        __syncthreads();                                           // also: the previous tile's output stage has read keys / idOf
        VP_FT_STAMP(1);
        // every entry of the list proposes the seed that sits at slot q (its coordinates are those of q) from slot s
        auto scatter = [&](uint32_t nlist, auto step) {
            constexpr int STEP = decltype(step)::value;
            auto run = [&](const auto& pos) {
                for (uint32_t e = tid; e < nlist; e += NT) {
                    const uint32_t entry = list[e], s = entry & 0xFFFFu, q = entry >> 16;
                    const float sx = pos.x((q / XR) & 3u, q % XR), sy = pos.y((q / (4u * XR)) & 3u), sz = pos.z(q / (16u * XR));
                    if constexpr (STEP == 2 && VP_PROPOSE_HALF) propose_half<XR>(keys, s, sx, sy, sz, pos);
                    else propose<XR, STEP>(keys, s, sx, sy, sz, pos, q);
                }
            };
            if constexpr (VP_FT_POS_LDS) run(ChainPosLds<XR>{posX, posY, posZ}); else run(ChainPosCalc{f, rx0, ry, rz, k});
        };
        // ---- stage A: border voxels -> pass with k = n/2
        // (Measured and dropped, profiles/r03/ab_gather_*.txt: stage A as a GATHER -- the wave ballots of the border flags in LDS, "some
        // candidate of this voxel is a border voxel" as an OR of four ballot words, the <= 8 candidates evaluated by the thread that
        // proposes the voxel in stage B; no list, scatter, collect or key reset and two barriers fewer.  Bit-identical, 0.302 -> 0.300 ms
        // at n = 512, 2.12 -> 2.32 ms at n = 1024: what it adds to the one proposing wave outweighs what it takes from the others.
        // The opposite trade -- three waves per 64 entries of stage B, one target plane each, a third of the instruction stream per wave
        // but the set-up three times -- ran 0.305 -> 0.344 ms / 2.13 -> 2.27 ms (ab_split_*.txt): total issue decides, not the critical path.)
        bool flag[PER];
        uint32_t seed[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t s = tid + (uint32_t)i * NT;
            const uint32_t rp = rpb + (uint32_t)i * G, y = ry + (rp & 3u) * k, z = rz + (rp >> 2) * k;
            flag[i] = xin && ((mw[u][i] >> (myx & 31u)) & 1u);
            keys[s] = kEmpty;
            idOf[s] = ID::pack(myx, y, z);
            seed[i] = s;                                           // a border voxel is its own seed
        }
        VP_FT_STAMP(2);                                            // border words arrived (first tile of the workgroup), keys / ids written
        append(flag, seed, cnt[0]);
        VP_FT_STAMP(3);
        __syncthreads();
        VP_FT_STAMP(4);
        scatter(cnt[0], std::integral_constant<int, 2>{});
        VP_FT_STAMP(5);
        __syncthreads();
        VP_FT_STAMP(6);
        // ---- stage B: voxels that have a seed now -> pass with k = n/4
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t s = tid + (uint32_t)i * NT;
            const unsigned long long key = keys[s];
            flag[i] = key != kEmpty;
            seed[i] = (uint32_t)key & 0xFFFFu;                     // slot of the voxel's pass-1 seed (garbage where flag is false: not appended)
            keys[s] = kEmpty;                                      // own slots only: nobody else touches them before the barrier
        }
        VP_FT_STAMP(7);
        append(flag, seed, cnt[1]);
        VP_FT_STAMP(8);
        __syncthreads();
        VP_FT_STAMP(9);
        scatter(cnt[1], std::integral_constant<int, 1>{});
        VP_FT_STAMP(10);
        __syncthreads();
        VP_FT_STAMP(11);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (!xin) continue;
            const unsigned long long key = keys[tid + (uint32_t)i * NT];
            const T id = key == kEmpty ? ID::none() : idOf[(uint32_t)key & 0x07FFFFFFu];       // stage B tags its proposals with the seed's slot
            // (Answering the lattices of at most one border voxel from the census in THESE tiles as well -- two thirds of their border voxels --
            // was measured: -1 % at n = 512, +4 % at n = 1024, profiles/r04/ab_ft3_*.txt: such a tile is bound by its fixed stages, not by
            // the number of proposals.)
            const uint32_t rp = rpb + (uint32_t)i * G;
            const size_t vox = (size_t)((rz + (rp >> 2) * k) * N + (ry + (rp & 3u) * k)) * N + myx;
            if constexpr (CPT) {
                const uint2 c = IdC::from64(id);
                ft_store(reinterpret_cast<uint32_t*>(out) + vox, c.x);
                ft_store(reinterpret_cast<unsigned char*>(reinterpret_cast<uint32_t*>(out) + (size_t)N * N * N) + vox, (unsigned char)c.y);
            } else {
                ft_store(out + vox, id);
            }
        }
        VP_FT_STAMP(12);
        VP_FT_FLUSH(tile);
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

#ifdef VP_ISA_PROBE    // dev: compile ONE kernel instance to look at its ISA (hipcc -DVP_ISA_PROBE='...' -S --cuda-device-only)
template __global__ void VP_ISA_PROBE;
}  // namespace
}  // namespace vp
#else
inline bool wide(const Frame& f) { return f.n > 1024; }           // 64-bit ids

}  // namespace

#if VP_PART_MAIN
size_t jfa_id_bytes(const Frame& f) { return wide(f) ? 8 : 4; }
#endif

// ---------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

#if VP_PART_MAIN
int launch_jfa_init(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* below,
                    const uint32_t* above, void* d_ids, uint32_t* d_border_words)
{
    const size_t nwords = (size_t)f.n * f.n * (f.z1 - f.z0) / 32;
    const unsigned blocks = (unsigned)(nwords / 256);             // nwords is a multiple of 256
    ProfScope p(ctx, d_ids ? VP_K_JFA_INIT : VP_K_SURFACE);
    static const int march = env_int("VP_BORDER_MARCH", 1);       // dev switch: 0 = jfa_init for the mask as well
    if (!d_ids && d_border_words && march && f.w <= 64) {
        // border mask alone: lanes march along z (jfa_border_march); chunks of zc planes, short enough to fill the chip
        const uint32_t rowsPerWave = 64u / f.w, wavesPerPlane = (f.n + rowsPerWave - 1) / rowsPerWave;
        const uint32_t inPlane = (wavesPerPlane + 3u) / 4u, nz = f.z1 - f.z0;
        uint32_t zc = 32;
        while (zc > 4 && inPlane * ((nz + zc - 1) / zc) < 8u * (uint32_t)ctx->cus) zc /= 2;
        hipLaunchKernelGGL(jfa_border_march, dim3(inPlane, (nz + zc - 1) / zc), dim3(256), 0, ctx->stream, f, d_words, below, above, d_border_words, zc);
        VP_HIP(hipGetLastError());
        return 0;
    }
#define VP_INIT(ID, I, M) hipLaunchKernelGGL((jfa_init<ID, I, M>), dim3(blocks), dim3(256), 0, ctx->stream, f, d_words, below, above, \
                                             (typename ID::T*)d_ids, d_border_words)
    if (wide(f)) {
        if (d_ids && d_border_words) VP_INIT(Id64, true, true); else if (d_ids) VP_INIT(Id64, true, false); else VP_INIT(Id64, false, true);
    } else {
        if (f.n <= 512) { if (d_ids && d_border_words) VP_INIT(Id9, true, true); else if (d_ids) VP_INIT(Id9, true, false); else VP_INIT(Id9, false, true); }
        else            { if (d_ids && d_border_words) VP_INIT(Id10, true, true); else if (d_ids) VP_INIT(Id10, true, false); else VP_INIT(Id10, false, true); }
    }
#undef VP_INIT
    VP_HIP(hipGetLastError());
    return 0;
}

int launch_jfa_pass(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                    const void* d_plus, void* d_out, int algo)
{
    return launch_jfa_pass_ex(ctx, f, k, d_in, d_minus, d_plus, d_out, algo, nullptr, 0.0f, nullptr);
}

bool jfa_can_start_from_mask(const Frame& f, int algo) { return algo == VP_ALGO_TILED && f.n >= VP_TILE_MIN_N && f.n % 128 == 0; }

// Compact id state (IdC) for the whole-grid sequence of vp_jfa at n > 1024: first two passes fused from the border mask, every later
// pass on the tile kernel -- all buffers are the call's own workspace, so the layout never meets a caller (the slab entry points keep
// the 8-byte ids vp_jfa_id_bytes reports).  VP_JFA_COMPACT=0 (dev / tests) keeps the 8-byte ids here too.
bool jfa_compact_applies(const Frame& f, int algo)
{
    static const int enabled = env_int("VP_JFA_COMPACT", 1);
    return enabled && wide(f) && jfa_can_fuse_first_two(f, algo);
}

// Passes k = n/2 and k = n/4 of a whole grid from its border mask in one launch (jfa_first_two); timed as the first pass.
#ifndef VP_JFA_FIRST_TWO_DEFAULT
#define VP_JFA_FIRST_TWO_DEFAULT 1
#endif
// The chains {r, r + n/4, r + n/2, r + 3n/4} are closed for any n % 4 == 0 (every legal n), and a tile whose 16 / 32 residues reach past n/4
// masks the excess lanes: the fused start serves EVERY whole grid the tile kernels serve.  (Until late in round 4 it borrowed the n % 128 == 0
// of jfa_first_pass -- the one-pass form the slab pipelines fall back to, jfa_can_start_from_mask.)  Measured (tools/size_sweep.sh,
// profiles/r04/size_sweep.txt): whole step 2.89 -> 2.47 ms at n = 480, 5.95 -> 4.55 at 544, 29.3 -> 23.6 at 960.
#ifndef VP_FIRST_TWO_ANY_N
#define VP_FIRST_TWO_ANY_N 1
#endif
bool jfa_can_fuse_first_two(const Frame& f, int algo)
{
    static const int enabled = env_int("VP_JFA_FIRST_TWO", VP_JFA_FIRST_TWO_DEFAULT);
    return enabled && algo == VP_ALGO_TILED && f.n >= VP_TILE_MIN_N && (VP_FIRST_TWO_ANY_N || f.n % 128 == 0) && f.z0 == 0 && f.z1 == f.n;
}

#ifndef VP_FIRST_TWO_TPW
#define VP_FIRST_TWO_TPW 2        // tiles per workgroup.  Round 3, without the census fast path (profiles/r03/ab_tpw_*.txt): 1 / 2 / 4 / 8 tiles = 0.371 / 0.373 / 0.370 / 0.406 ms
                                  // at n = 512, 2.78 / 2.81 / 2.74 / 3.23 at n = 1024 -- the other resident workgroups already cover the load
                                  // latency; what did pay is the ONE-dimensional launch these forms share: 0.405 -> 0.371 ms, 2.92 -> 2.78
#endif
int launch_jfa_first_two(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, void* d_out)
{
    ProfScope p(ctx, VP_K_JFA_FIRST);
    const uint32_t k = f.n / 4;
    // tile = 4 x 4 x 4 chain positions x XR residues.  Measured (profiles/r02/ab33.txt): 16 residues x 256 threads is the best
    // shape at n = 512 (0.404 ms against 0.183 + 0.307 for the two separate kernels; 32 x 512: 0.436), 32 x 512 at n = 1024
    // (2.94 ms against 1.39 + 2.12; 16 x 256: 3.41).  A workgroup takes VP_FIRST_TWO_TPW consecutive tiles.
#ifndef VP_FT_SMALL_XR32
#define VP_FT_SMALL_XR32 0        // 32 residues x 512 threads at n <= 512 too (128-byte row segments instead of 64): 0.238 -> 0.297 ms with the census
                                  // (fewer tiles without a contested lattice), profiles/r04/ab_ftxr_*.txt
#endif
#ifndef VP_FT_BIG_XR16
#define VP_FT_BIG_XR16 0          // 16 residues x 256 threads at n = 1024 too: 1.60 -> 1.69 ms (profiles/r04/ab_ftxr_*.txt)
#endif
    const bool small = (f.n <= 512 && !VP_FT_SMALL_XR32) || (VP_FT_BIG_XR16 && f.n <= 1024);
    const uint32_t xr = small ? 16u : 32u;
    const uint32_t tilesX = (k + xr - 1) / xr, tiles = tilesX * k * k;
    const dim3 grid((tiles + VP_FIRST_TWO_TPW - 1) / VP_FIRST_TWO_TPW);
    auto pow2 = [](uint32_t v) { return v != 0 && (v & (v - 1)) == 0; };
    const uint32_t shifts = (pow2(tilesX) && pow2(k)) ? ((uint32_t)__builtin_ctz(tilesX) | ((uint32_t)__builtin_ctz(k) << 8) | (1u << 16)) : 0u;
    if (wide(f) && f.compact) hipLaunchKernelGGL((jfa_first_two<Id64, 32, 512, VP_FIRST_TWO_TPW, true>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint2*)d_out, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k));
    else if (wide(f)) hipLaunchKernelGGL((jfa_first_two<Id64, 32, 512, VP_FIRST_TWO_TPW>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint2*)d_out, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k));
    else if (small && f.n <= 512) hipLaunchKernelGGL((jfa_first_two<Id9, 16, 256, VP_FIRST_TWO_TPW>), grid, dim3(256), 0, ctx->stream, f, d_border, (uint32_t*)d_out, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k));
#if VP_FT_BIG_XR16
    else if (small) hipLaunchKernelGGL((jfa_first_two<Id10, 16, 256, VP_FIRST_TWO_TPW>), grid, dim3(256), 0, ctx->stream, f, d_border, (uint32_t*)d_out, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k));
#endif
#if VP_FT_SMALL_XR32
    else if (f.n <= 512) hipLaunchKernelGGL((jfa_first_two<Id9, 32, 512, VP_FIRST_TWO_TPW>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint32_t*)d_out, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k));
#endif
    else            hipLaunchKernelGGL((jfa_first_two<Id10, 32, 512, VP_FIRST_TWO_TPW>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint32_t*)d_out, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k));
    VP_HIP(hipGetLastError());
    return 0;
}

// First pass from the whole-grid border mask (see jfa_first_pass).
int launch_jfa_first_pass(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, void* d_out)
{
    ProfScope p(ctx, VP_K_JFA_FIRST);
    const uint32_t gx = (f.n + 255) / 256, gy = f.n / kFirstRows;
    const dim3 grid(gx * gy * (f.z1 - f.z0));
    if (wide(f)) hipLaunchKernelGGL(jfa_first_pass<Id64>, grid, dim3(256), 0, ctx->stream, f, f.n / 2, d_border, (uint2*)d_out, gx, gy);
    else if (f.n <= 512) hipLaunchKernelGGL(jfa_first_pass<Id9>, grid, dim3(256), 0, ctx->stream, f, f.n / 2, d_border, (uint32_t*)d_out, gx, gy);
    else         hipLaunchKernelGGL(jfa_first_pass<Id10>, grid, dim3(256), 0, ctx->stream, f, f.n / 2, d_border, (uint32_t*)d_out, gx, gy);
    VP_HIP(hipGetLastError());
    return 0;
}

bool jfa_pass_can_fuse_final(const Frame& f, uint32_t k, int algo)
{
    (void)k;
    return algo == VP_ALGO_TILED && f.n >= VP_TILE_MIN_N;
}
#endif  // VP_PART_MAIN

// One row of "none" per id format for out-of-grid reads: 2048 x 8 bytes of the 64-bit "none", then 1024 x 4 of each 32-bit one, then
// the compact format's word row (2048 x 4) with its byte row (2048 x 1) right behind it.
static int ensure_none_rows(vp_ctx* ctx)
{
    if (ctx->none_row.ptr) return 0;
    VP_TRY(reserve(ctx, ctx->none_row, 2048 * 8 + 2 * 1024 * 4 + 2048 * 4 + 2048));
    char* p = (char*)ctx->none_row.ptr;
    VP_HIP(hipMemsetAsync(p, 0xFF, 2048 * 8, ctx->stream));
    VP_HIP(hipMemsetD32Async((hipDeviceptr_t)(p + 2048 * 8), (int)kNone9, 1024, ctx->stream));
    VP_HIP(hipMemsetD32Async((hipDeviceptr_t)(p + 2048 * 8 + 1024 * 4), (int)kNone10, 1024, ctx->stream));
    VP_HIP(hipMemsetD32Async((hipDeviceptr_t)(p + 2048 * 8 + 2 * 1024 * 4), (int)IdC::kNoneWord, 2048, ctx->stream));
    VP_HIP(hipMemsetAsync(p + 2048 * 8 + 2 * 1024 * 4 + 2048 * 4, (int)IdC::kNoneByte, 2048, ctx->stream));
    return 0;
}
template <class ID>
static const typename ID::T* none_row_of(vp_ctx* ctx)
{
    const char* p = (const char*)ctx->none_row.ptr;
    if (std::is_same<ID, Id9>::value) p += 2048 * 8;
    if (std::is_same<ID, Id10>::value) p += 2048 * 8 + 1024 * 4;
    if (std::is_same<ID, IdC>::value) p += 2048 * 8 + 2 * 1024 * 4;
    return (const typename ID::T*)p;
}

template <class ID>
static int launch_chain(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus, const void* d_plus,
                        void* d_out, const uint32_t* d_words, float fill, float* d_sdf)
{
    using T = typename ID::T;
    const uint32_t nz = f.z1 - f.z0;
    VP_TRY(ensure_none_rows(ctx));
    const T* none_row = none_row_of<ID>(ctx);
    const bool skip = k * 4 >= f.n, fin = d_sdf != nullptr;
    const uint32_t nres = std::min(k, nz), zlen = (nz + k - 1) / k; // residue classes of the local plane index, planes per class
    const uint32_t nresY = std::min(k, f.n), ylen = (f.n + k - 1) / k;
    const bool closedZ = f.z0 == 0 && f.z1 == f.n && f.n == 8u * k;         // plane chains of exactly eight: no halo planes (CZ)
    // Tile = RY rows x CH planes per workgroup and the table size.
#define VP_LAUNCH_CHAIN(TAB, PXT, RY, CH, S, C, F) VP_LAUNCH_CHAIN_CZ(TAB, PXT, RY, CH, S, C, F, false)
#define VP_LAUNCH_CHAIN_CZ(TAB, PXT, RY, CH, S, C, F, Z)                                                                            \
    hipLaunchKernelGGL((jfa_pass_zstream<ID, TAB, PXT, RY, CH, S, C, F, Z>),                                                            \
                       VP_ZSTREAM_1D ? dim3(nresY * ((ylen + RY - 1) / RY) * nres * ((zlen + CH - 1) / CH)) : dim3(nresY * ((ylen + RY - 1) / RY), nres * ((zlen + CH - 1) / CH)), dim3(256), 0, ctx->stream, f, k,         \
                       (const T*)d_in, (const T*)d_minus, (const T*)d_plus, (T*)d_out, none_row, d_words, fill, d_sdf, nresY * ((ylen + RY - 1) / RY))
    // 64-bit ids: chk = n == table size, "none" has no spare table slot -> explicit test.  32-bit ids (U): the x table has one
    // more slot (+inf, the x index of "none"), "none" needs no test; the sparse variant keeps the flag all the same (most ids
    // are "none" there and the flag skips their updates: 0.355 vs 0.377 ms, round 1).
#define VP_LAUNCH_TILE(TAB, U, RY, CH, CHD)                                                                                          \
    do {                                                                                                                             \
        const bool chk = !U && (int)f.n >= TAB;                                                                                      \
        const bool deep = zlen % CHD == 0;                        /* dense / last pass: longer plane chains when they divide evenly */ \
        constexpr int PXW = U ? TAB + 1 : TAB;                                                                                       \
        if (skip)      { if (chk || U) VP_LAUNCH_CHAIN(TAB, PXW, RY, CH, true, true, false);  else VP_LAUNCH_CHAIN(TAB, PXW, RY, CH, true, false, false); }  \
        else if (fin && deep) { if (chk) VP_LAUNCH_CHAIN(TAB, PXW, RY, CHD, false, !U, true);  else VP_LAUNCH_CHAIN(TAB, PXW, RY, CHD, false, false, true); }  \
        else if (fin)  { if (chk) VP_LAUNCH_CHAIN(TAB, PXW, RY, CH, false, !U, true);   else VP_LAUNCH_CHAIN(TAB, PXW, RY, CH, false, false, true); }  \
        else if (deep && closedZ && CHD == 8) { if (chk) VP_LAUNCH_CHAIN_CZ(TAB, PXW, RY, CHD, false, !U, false, true); else VP_LAUNCH_CHAIN_CZ(TAB, PXW, RY, CHD, false, false, false, true); } \
        else if (deep) { if (chk) VP_LAUNCH_CHAIN(TAB, PXW, RY, CHD, false, !U, false); else VP_LAUNCH_CHAIN(TAB, PXW, RY, CHD, false, false, false); } \
        else           { if (chk) VP_LAUNCH_CHAIN(TAB, PXW, RY, CH, false, !U, false);  else VP_LAUNCH_CHAIN(TAB, PXW, RY, CH, false, false, false); } \
    } while (0)
    if constexpr (std::is_same<ID, Id64>::value) VP_LAUNCH_TILE(Id64::kTab, false, kRowsWide, kPlanesWide, kPlanesWideDense);
    else if constexpr (std::is_same<ID, Id9>::value) VP_LAUNCH_TILE(512, true, kRows, kPlanes, kPlanesDense);     // 2-KB tables
    else VP_LAUNCH_TILE(1024, true, kRows, kPlanes, kPlanes);     // 4-KB tables: 4x8 costs occupancy (4.93 vs 4.70 ms at n = 1024)
#undef VP_LAUNCH_TILE
#undef VP_LAUNCH_CHAIN
#undef VP_LAUNCH_CHAIN_CZ
    return 0;
}

// Dense tile kernel (jfa_pass_dense): 32-bit ids, dense passes (and, as build options, the wide passes and the fused last
// pass), id buffers contiguous.

#ifndef VP_JFA_DENSE_FINAL
#define VP_JFA_DENSE_FINAL 1      // the fused last pass runs here too: 0.397 -> 0.368 ms at n = 512, 3.62 -> 3.42 ms at n = 1024 against
                                  // jfa_pass_zstream<FINAL> (profiles/r02/ab19.txt; it was 0 - 5 % behind before the IdU ids, the pipelined lookups and tail_split)
#endif
#ifndef VP_JFA_DENSE_WIDEK
#define VP_JFA_DENSE_WIDEK 0      // 1: route the wide passes (k >= n/4) here too (SKIP form; measured 5 - 9 % behind: they wait on
#endif                            //    scattered row segments, not on VALU issue)
#ifndef VP_JFA_DENSE_DEFAULT
#define VP_JFA_DENSE_DEFAULT 1
#endif

static bool dense_applies(const Frame& f, uint32_t k, const void* d_in, const void* d_minus, const void* d_plus, bool fin)
{
    if (fin && !VP_JFA_DENSE_FINAL) return false;
    static const int enabled = env_int("VP_JFA_DENSE", VP_JFA_DENSE_DEFAULT);         // dev switch: 0 = round-1 kernel for every pass
    // 8-byte ids (n > 1024): the fused last pass runs here (38.4 -> 37.7 ms at n = 2048); the id passes stay on jfa_pass_zstream
    // (46.5 ms against 51.2 here, profiles/r03/n2048_dense_wide.txt): the winner gather -- one 8-byte load per voxel from rows the tile
    // read one to three planes earlier -- misses the L2 at this size (16-KB rows; 43.1 ms with the gather ablated).
    // VP_JFA_DENSE_WIDE=1 routes them here all the same (dev), =0 keeps even the last pass on the round-1 kernel.
    static const int wideMode = env_int("VP_JFA_DENSE_WIDE", 2);
    if (f.compact) return true;                                    // compact ids (whole-grid vp_jfa at n > 1024): every pass k < n/4 runs here
    if (wide(f) && (wideMode == 0 || (wideMode == 2 && !fin))) return false;
    if (!enabled || (k * 4 >= f.n && !VP_JFA_DENSE_WIDEK)) return false;
    const size_t plane = (size_t)f.n * f.n * jfa_id_bytes(f);
    const char* in = (const char*)d_in;
    if (f.z0 > 0 && (const char*)d_minus + (size_t)k * plane != in) return false;      // the three id buffers must be one volume
    const uint32_t pbase = std::max(f.z1, f.z0 + k);
    if (f.z1 < f.n && (const char*)d_plus != in + (size_t)(pbase - f.z0) * plane) return false;
    return true;
}

// Tail of a dense launch.  A launch of T tiles on S = CUs x workgroups-per-CU slots runs ~T/S rounds; at n = 512 that is 5.3:
// while the chip drains, slots stand empty for about half a tile time (77 us of a 410-us pass).  The last ~4/3 S tiles are
// therefore dispatched as two half-row units each (x halves, one table prologue more per split tile): -1.9 % on the dense passes
// at n = 512 (profiles/r02/ab17.txt, ab18.txt).  Launches of 16 rounds and more (n = 1024: 43) are left whole (measured +-0).
static uint32_t tail_split(const vp_ctx* ctx, uint32_t tiles, uint32_t wgPerCu)
{
    static const int forced = env_int("VP_DENSE_SPLIT", -1);       // dev switch: number of tiles to split
    if (forced >= 0) return std::min(tiles, (uint32_t)forced);
    const uint32_t slots = (uint32_t)ctx->cus * wgPerCu;
    if (tiles >= 16u * slots) return 0;
    return std::min(tiles / 2u, slots * 4u / 3u);
}

// The seed-scatter kernel serves the pass with k = n/4 of a whole grid (the chains must be closed: 4 k = n), either id width.
#ifndef VP_JFA_SEEDS_DEFAULT
#define VP_JFA_SEEDS_DEFAULT 1
#endif
static bool seeds_applies(const Frame& f, uint32_t k, bool fin)
{
    static const int enabled = env_int("VP_JFA_SEEDS", VP_JFA_SEEDS_DEFAULT);
    return enabled && !fin && !f.compact && k * 4u == f.n && f.z0 == 0 && f.z1 == f.n && k <= 65535u;
}

template <class ID>
static int launch_dense(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf)
{
    const uint32_t nz = f.z1 - f.z0;
    VP_TRY(ensure_none_rows(ctx));
    using T = typename ID::T;
    const T* none_row = none_row_of<ID>(ctx);
    const bool fin = d_sdf != nullptr;
    const bool wideK = k * 4 >= f.n;                               // half of the neighbour rows / planes are outside the grid: SKIP variant
    const uint32_t nres = std::min(k, nz), zlen = (nz + k - 1) / k;
    const uint32_t nresY = std::min(k, f.n), ylen = (f.n + k - 1) / k;
    // Plane chains that are not a multiple of eight (the sub-slab regions of the multi-GPU pipelines: 288 planes at k = 32 are chains
    // of nine).  A tile of CH planes walks CH + 2 plane iterations whatever it outputs, so 4-plane tiles -- what round 2 ran on every
    // such chain -- cost 6 per 4 planes.  The first 8 q members of every chain are the CONTIGUOUS planes [z0, z0 + 8 q k): they go to
    // the 8-plane form as a sub-slab of their own (the id volume is contiguous, so each part sees the other as its halo), the remaining
    // r < 8 members as a second launch with whichever tile is cheaper for r (10 iterations for one 8-tile, 6 per 4-tile).
    // Measured per rank on one GPU (tools/ghost_prof.py, tools/slab_scaling.py): see DESIGN.md section 6.
#ifndef VP_DENSE_SPLIT_CHAINS
#define VP_DENSE_SPLIT_CHAINS 1
#endif
    if (VP_DENSE_SPLIT_CHAINS && !dense_wide<ID>() && nz % k == 0 && k < nz && zlen > 8 && zlen % 8 != 0) {
        const uint32_t planesA = (zlen / 8u) * 8u * k;
        Frame fa = f, fb = f;
        fa.z1 = f.z0 + planesA; fb.z0 = fa.z1;
        const size_t idPlane = (size_t)f.n * f.n * jfa_id_bytes(f), wordPlane = (size_t)f.n * f.w, sdfPlane = (size_t)f.n * f.n;
        VP_TRY(launch_dense<ID>(ctx, fa, k, d_in, d_out, d_words, fill, d_sdf));
        return launch_dense<ID>(ctx, fb, k, (const char*)d_in + planesA * idPlane, (char*)d_out + planesA * idPlane,
                                d_words ? d_words + planesA * wordPlane : nullptr, fill, d_sdf ? d_sdf + planesA * sdfPlane : nullptr);
    }
    // Tile 4 rows x 8 planes when the plane chains divide by 8, else 4 x 4.  2-KB tables (n <= 512): 256 threads, 26 KB of LDS,
    // six workgroups per CU.  4-KB tables: the 4 x 8 tile takes 52 KB, shared by the 8 waves of a 512-thread workgroup (three per CU).
#ifndef VP_DENSE_WIDE_NT
#define VP_DENSE_WIDE_NT 512      // threads per workgroup with 8-byte ids
#endif
#ifndef VP_DENSE_RY
#define VP_DENSE_RY 4             // output rows per tile
#endif
#ifndef VP_DENSE_RY_SMALL
#define VP_DENSE_RY_SMALL 8       // ... of the id passes with the 2-KB tables (n <= 512): 8 rows = 109 VGPRs, four waves per SIMD, but half the tiles -- half the
#endif                            // table builds and tile prologues -- and 3.1 instead of 3.75 decoded ids per voxel: dense -1.1 % at n = 512; with the 4-KB tables
                                  // +1.1 %, the fused last pass +1.1 % / +5 % (profiles/r03/ab_ry8_*.txt): those keep 4 rows
#ifndef VP_DENSE_K2_PLAIN
#define VP_DENSE_K2_PLAIN 1
#endif
#ifndef VP_DENSE_RY8_1024
#define VP_DENSE_RY8_1024 0       // 8-row tiles for the id passes with the 4-KB tables (68 KB of LDS, four waves per SIMD, with FULL): +-0 (21.72 vs
                                  // 21.74 ms over the seven dense passes at n = 1024, profiles/r04/ab_gnt_ry8_1024.txt)
#endif
#ifndef VP_DENSE_FULL
#define VP_DENSE_FULL 1           // compile-time row / plane counts where every tile is whole (see jfa_pass_dense)
#endif
#ifndef VP_DENSE_PAIRS_DEFAULT
#define VP_DENSE_PAIRS_DEFAULT 2  // pair mode (see jfa_pass_dense): 0 off, 1 the fused last pass only, 2 every dense pass too (profiles/r03/ab_pairs_*.txt)
#endif
    static const int pairs = env_int("VP_DENSE_PAIRS", VP_DENSE_PAIRS_DEFAULT);
    const bool pow2 = (f.n & (f.n - 1)) == 0;
#define VP_LAUNCH_DENSE_PM(CH, NT, F, S, PM)                                                                                       \
    do {                                                                                                                           \
        constexpr int RY_ = (ID::kTab == 512 && !(F) && !(S) && CH == 8) ? VP_DENSE_RY_SMALL : (VP_DENSE_RY8_1024 && ID::kTab == 1024 && !(F) && !(S) && CH == 8) ? 8 : VP_DENSE_RY;  \
        const uint32_t ty_ = nresY * ((ylen + RY_ - 1) / RY_), t_ = ty_ * nres * ((zlen + CH - 1) / CH);                           \
        /* a row of <= NT voxels has no halves */                                                                                  \
        const uint32_t sp_ = f.n > NT ? tail_split(ctx, t_, dense_wide<ID>() ? (NT == 256 ? 5u : 2u) : ID::kTab == 512 ? (RY_ == 8 ? 4u : (F) && !final_mask_global<ID>() ? 5u : 6u) : NT == 512 ? ((F) && !final_mask_global<ID>() ? 2u : 3u) : 4u) : 0u;                      \
        /* FULL tiles: compile-time row / plane counts (only the 8-plane forms of whole chains get the second instantiation) */    \
        /* (measured, profiles/r04/ab_full_*.txt: dense -3.2 % at n = 512, -5 % at n = 2048, fused last pass -1 / -4 / -3 %; the ID PASSES */ \
        /* with the 4-KB tables lose 4 %: 80 VGPRs + 8 spilled instead of 71 under the six-wave bound -- those keep the run-time counts)    */ \
        constexpr bool fullOk_ = CH == 8 && !(S) && (ID::kTab != 1024 || (F) || VP_DENSE_FULL_1024);                               \
        const bool full_ = VP_DENSE_FULL && fullOk_ && ylen % RY_ == 0 && zlen % CH == 0 && nz % k == 0 && f.n % k == 0;           \
        if (full_) hipLaunchKernelGGL((jfa_pass_dense<ID, RY_, CH, NT, F, true, S, PM, 0, fullOk_>), dim3(t_ + sp_), dim3(NT), 0, ctx->stream, f, k, \
                           (const T*)d_in, (T*)d_out, none_row, d_words, fill, d_sdf, ty_, t_, sp_);                               \
        else hipLaunchKernelGGL((jfa_pass_dense<ID, RY_, CH, NT, F, true, S, PM>), dim3(t_ + sp_), dim3(NT), 0, ctx->stream, f, k,                \
                           (const T*)d_in, (T*)d_out, none_row, d_words, fill, d_sdf, ty_, t_, sp_);                               \
    } while (0)
    // pair mode where it applies: 32-bit ids, n a power of two, whole rows of NT-thread iterations; the lane permutation follows k
#define VP_LAUNCH_DENSE(CH, NT, F, S)                                                                                              \
    do {                                                                                                                           \
        if constexpr ((!dense_wide<ID>() || std::is_same<ID, IdC>::value) && !(S) && CH == 8) {    /* the 4-plane tiles of shallow slabs keep the plain form (fewer kernels to build) */ \
            const bool pm_ = pow2 && f.n % NT == 0 && pairs >= ((F) ? 1 : 2) && ((F) || k >= 2);                                   \
            if (pm_ && (F))        { VP_LAUNCH_DENSE_PM(CH, NT, F, S, 1); break; }                                                 \
            /* k = 2 with the 2-KB tables: the plain form is 7 % faster (0.370 vs 0.396 ms; with the 4-KB tables pairs win by 3 %) */ \
            if (pm_ && k == 2 && ID::kTab == 512 && VP_DENSE_K2_PLAIN) { VP_LAUNCH_DENSE_PM(CH, NT, F, S, 0); break; }                                  \
            if (pm_ && k == 2)     { VP_LAUNCH_DENSE_PM(CH, NT, F, S, 2); break; }                                                 \
            if (pm_ && k == 4)     { VP_LAUNCH_DENSE_PM(CH, NT, F, S, 4); break; }                                                 \
            if (pm_)               { VP_LAUNCH_DENSE_PM(CH, NT, F, S, 8); break; }                                                 \
        }                                                                                                                          \
        VP_LAUNCH_DENSE_PM(CH, NT, F, S, 0);                                                                                       \
    } while (0)
#define VP_DENSE_F(CH, NT) do { if (fin) VP_LAUNCH_DENSE(CH, NT, (VP_JFA_DENSE_FINAL != 0), false);                                \
                                else if (wideK) VP_LAUNCH_DENSE(CH, NT, false, (VP_JFA_DENSE_WIDEK != 0));                          \
                                else VP_LAUNCH_DENSE(CH, NT, false, false); } while (0)
    // one 8-plane tile per chain also where the chain has 5 .. 7 members (10 plane iterations against 2 x 6)
    const bool deep = zlen % 8 == 0 || (VP_DENSE_SPLIT_CHAINS && !dense_wide<ID>() && zlen > 4 && zlen < 8);
#ifndef VP_DENSE_CLOSED
#define VP_DENSE_CLOSED 1         // closed tiles at k = n/8 (see jfa_pass_dense)
#endif
    if constexpr (!dense_wide<ID>()) {
        constexpr uint32_t NTC = ID::kTab == 512 ? 256u : 512u;
        if (VP_DENSE_CLOSED && pairs >= 2 && pow2 && !fin && f.z0 == 0 && f.z1 == f.n && f.n == 8u * k && f.n % NTC == 0) {
#ifndef VP_DENSE_CLOSED_ROWS
#define VP_DENSE_CLOSED_ROWS 1    // close the rows as well (8 x 8 tiles); 0 = planes only (4 x 8 tiles)
#endif
            constexpr int RYC = VP_DENSE_CLOSED_ROWS ? 8 : 4, CL = VP_DENSE_CLOSED_ROWS ? 3 : 2;
            const uint32_t ty_ = nresY * ((ylen + RYC - 1) / RYC), t_ = ty_ * nres;          // one plane chain per tile
            const uint32_t sp_ = f.n > NTC ? tail_split(ctx, t_, ID::kTab == 512 ? (RYC == 8 ? 4u : 6u) : (RYC == 8 ? 2u : 3u)) : 0u;
            hipLaunchKernelGGL((jfa_pass_dense<ID, RYC, 8, NTC, false, true, false, 8, CL, (VP_DENSE_FULL != 0 && ID::kTab == 512)>), dim3(t_ + sp_), dim3(NTC), 0, ctx->stream, f, k,
                               (const T*)d_in, (T*)d_out, none_row, d_words, fill, d_sdf, ty_, t_, sp_);
            return 0;
        }
    }
    // 8-byte ids: 8-KB tables (PX + 4 x TY + one z position table = 48 KB), 512 threads
    // chains of one or two planes (the remainders of the split above are mostly that): 2-plane tiles, 4 plane iterations instead of 6
    const bool tiny = VP_DENSE_SPLIT_CHAINS && !fin && !wideK && zlen <= 2;
    if constexpr (dense_wide<ID>()) { if (deep) VP_DENSE_F(8, VP_DENSE_WIDE_NT); else VP_DENSE_F(4, VP_DENSE_WIDE_NT); }
    else if constexpr (ID::kTab == 512) { if (deep) VP_DENSE_F(8, 256); else if (tiny) VP_LAUNCH_DENSE(2, 256, false, false); else VP_DENSE_F(4, 256); }
    else                           { if (deep) VP_DENSE_F(8, 512); else if (tiny) VP_LAUNCH_DENSE(2, 256, false, false); else VP_DENSE_F(4, 256); }
#undef VP_DENSE_F
#undef VP_LAUNCH_DENSE
#undef VP_LAUNCH_DENSE_PM
    return 0;
}

// launch_dense<ID> per id format, one build part each (see VP_JFA_PART)
int launch_dense_id9(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf);
int launch_dense_id10(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf);
int launch_dense_idc(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf);
int launch_dense_id64(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf);
#if VP_PART_HAS(1)
int launch_dense_id9(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf)
{ return launch_dense<Id9>(ctx, f, k, d_in, d_out, d_words, fill, d_sdf); }
#endif
#if VP_PART_HAS(2)
int launch_dense_id10(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf)
{ return launch_dense<Id10>(ctx, f, k, d_in, d_out, d_words, fill, d_sdf); }
#endif
#if VP_PART_HAS(3)
int launch_dense_idc(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf)
{ return launch_dense<IdC>(ctx, f, k, d_in, d_out, d_words, fill, d_sdf); }
#endif
#if VP_PART_HAS(4)
int launch_dense_id64(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, void* d_out, const uint32_t* d_words, float fill, float* d_sdf)
{ return launch_dense<Id64>(ctx, f, k, d_in, d_out, d_words, fill, d_sdf); }
#endif

#if VP_PART_MAIN
// d_sdf != nullptr: this is the last pass and it writes the sdf directly (only where
// jfa_pass_can_fuse_final() says so); otherwise ids go to d_out.
int launch_jfa_pass_ex(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                       const void* d_plus, void* d_out, int algo, const uint32_t* d_words, float fill, float* d_sdf)
{
    const uint32_t nz = f.z1 - f.z0;
    // timing key: the tile kernels are reported per variant (their algorithmic bytes differ)
    const bool tile = algo != VP_ALGO_NAIVE && f.n >= VP_TILE_MIN_N;
    ProfScope p(ctx, !tile ? VP_K_JFA_PASS : d_sdf ? VP_K_JFA_LAST : k * 4 >= f.n ? VP_K_JFA_SPARSE : VP_K_JFA_DENSE);
    if (algo == VP_ALGO_NAIVE) {
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
    } else if (f.n >= VP_TILE_MIN_N && seeds_applies(f, k, d_sdf != nullptr)) {
        // tile = 4 x 4 x 4 chain positions x 32 residues = 2,048 voxels (28 KB of LDS), 512 threads = 4 voxels each.  Measured
        // (profiles/r02/ab28.txt, ab29.txt): 16 x 256 -14 %, 32 x 512 -16 % (n = 512) / -28 % (n = 1024) against
        // jfa_pass_zstream<SKIP>; 2 or 8 voxels per thread, 64-byte and 256-byte segments are all slower than that
#ifndef VP_SEEDS_XR
#define VP_SEEDS_XR 32
#define VP_SEEDS_NT 512
#endif
        const dim3 grid((k + VP_SEEDS_XR - 1) / VP_SEEDS_XR, k, k);  // 4 x 4 x 4 chain positions x XR residues per tile
        if (wide(f))         hipLaunchKernelGGL((jfa_pass_seeds<Id64, VP_SEEDS_XR, VP_SEEDS_NT>), grid, dim3(VP_SEEDS_NT), 0, ctx->stream, f, k, (const uint2*)d_in, (uint2*)d_out);
        else if (f.n <= 512) hipLaunchKernelGGL((jfa_pass_seeds<Id9, VP_SEEDS_XR, VP_SEEDS_NT>), grid, dim3(VP_SEEDS_NT), 0, ctx->stream, f, k, (const uint32_t*)d_in, (uint32_t*)d_out);
        else                 hipLaunchKernelGGL((jfa_pass_seeds<Id10, VP_SEEDS_XR, VP_SEEDS_NT>), grid, dim3(VP_SEEDS_NT), 0, ctx->stream, f, k, (const uint32_t*)d_in, (uint32_t*)d_out);
    } else if (f.n >= VP_TILE_MIN_N && dense_applies(f, k, d_in, d_minus, d_plus, d_sdf != nullptr)) {
        if (f.compact)       VP_TRY(launch_dense_idc(ctx, f, k, d_in, d_out, d_words, fill, d_sdf));
        else if (wide(f))    VP_TRY(launch_dense_id64(ctx, f, k, d_in, d_out, d_words, fill, d_sdf));
        else if (f.n <= 512) VP_TRY(launch_dense_id9(ctx, f, k, d_in, d_out, d_words, fill, d_sdf));
        else                 VP_TRY(launch_dense_id10(ctx, f, k, d_in, d_out, d_words, fill, d_sdf));
    } else if (f.n >= VP_TILE_MIN_N) {
        if (wide(f)) VP_TRY(launch_chain<Id64>(ctx, f, k, d_in, d_minus, d_plus, d_out, d_words, fill, d_sdf));
        else if (f.n <= 512) VP_TRY(launch_chain<Id9>(ctx, f, k, d_in, d_minus, d_plus, d_out, d_words, fill, d_sdf));
        else         VP_TRY(launch_chain<Id10>(ctx, f, k, d_in, d_minus, d_plus, d_out, d_words, fill, d_sdf));
    } else {
        const int RY = (int)(256 / f.n);
        const dim3 grid((f.n + RY - 1) / RY, nz);
        const size_t lds = (size_t)(2 + RY) * kTableKernelTab * sizeof(float);
        hipLaunchKernelGGL(jfa_pass_table, grid, dim3(256), lds, ctx->stream, f, k, (const uint32_t*)d_in, (const uint32_t*)d_minus,
                           (const uint32_t*)d_plus, (uint32_t*)d_out, RY);
    }
    VP_HIP(hipGetLastError());
    return 0;
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

#ifdef VP_FIRST_TWO_TIMING
extern "C" int vp_dev_first_two_times(unsigned long long* out16, int reset)
{
    static std::vector<unsigned long long> host((size_t)kFtSlots * 16);
    if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(g_ft_slots), host.size() * 8) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    // VP_FT_PHASE=p: only the tiles with tile % VP_FIRST_TWO_TPW == p (first, second, ... tile of a workgroup)
    const char* ph = getenv("VP_FT_PHASE");
    const long phase = ph ? atol(ph) : -1;
    for (size_t w = 0; w < kFtSlots; ++w)
        if (phase < 0 || (long)(w % VP_FIRST_TWO_TPW) == phase)
            for (int i = 0; i < 16; ++i) out16[i] += host[w * 16 + i];
    if (reset) { std::fill(host.begin(), host.end(), 0ull); if (hipMemcpyToSymbol(HIP_SYMBOL(g_ft_slots), host.data(), host.size() * 8) != hipSuccess) return 1; }
    return 0;
}
#endif

#endif  // VP_PART_MAIN

}  // namespace vp
#endif  // VP_ISA_PROBE
