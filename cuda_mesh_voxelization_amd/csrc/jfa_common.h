// jfa_common.h -- what the JFA kernels of libvphip.so share: id formats, the reference's float expressions, buffer-resource
// loads / stores, and the host-side description of an id WINDOW (include/vphip.h, vp_jfa_window_*).
//
// Result contract of every kernel built on this: the sdf of the reference's sequential JFA
// (/root/reference/vplib/src/jfa/sequential.cpp:7-127), bit for bit: same passes (k = n/2 .. 1, :72), same 26-neighbour scan order
// (z, y, x outer->inner, :86-88), strict '<' acceptance (:106), same float expressions for positions (:79-81) and distances
// (jfa/jfa.h:19-20), Jacobi update.
//
// State: the reference keeps float sdf + float3 seed position per voxel (16 B, two copies, plus a deep copy per pass, :123-124).
// Here the state is ONE packed id per voxel -- the voxel coordinates of the best seed so far.  The seed position and the distance are
// recomputed from it with the reference's expressions, which gives the same floats because the reference's stored sdf is itself the
// result of exactly that expression.  Built with -ffp-contract=off (an FMA changes the result).
#pragma once
#include "vp_internal.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

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
struct IdC {
    using T = uint2;
    static constexpr int kTab = 2048;
    static constexpr int kTabZ = 3072;                             // entries of the z table: 2048 real slots + 1024 of +inf for "none"
    static constexpr uint32_t kMask = 0x1FFCu;
    static constexpr uint32_t kNoneWord = 0xFFFFF800u;
    static constexpr uint32_t kNoneBit = 4u;                       // in the byte; bit 1 = top bit of scr(z)
    static constexpr uint32_t kNoneByte = 4u;
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
    __device__ static __forceinline__ uint32_t xoff(T a) { return (a.x & 0x7FFu) << 2; }
    // byte 2: slot 1024 + .., 4 ("none"): 2048 + ..  The byte is one of 0, 2, 4 in every plane a kernel of this library wrote or
    // vp_jfa_window_clear filled (a window is cleared whenever its geometry changes, so no other value is ever decoded).
    __device__ static __forceinline__ uint32_t zoff(T a) { return ((a.x >> 20) & 0xFFCu) | (a.y << 11); }
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

// ------------------------------------------------------------------------------------------ buffer-resource loads / stores
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
// An empty asm that "modifies" a running value: everything feeding it has to be computed here.  Without it the
// compiler sinks the compare/select chains of a whole chain towards the stores and keeps every distance live
// (132 VGPRs instead of 75).
__device__ __forceinline__ void pin(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(uint32_t& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(uint2& a) { asm volatile("" : "+v"(a.x), "+v"(a.y)); }

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

// lane i <- lane i-1 / lane i+1 of the wave (v_mov_b32_dpp wave_shr / wave_shl: a VALU move, not the ds_bpermute of __shfl)
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }   // lane i <- lane i-1
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }   // lane i <- lane i+1

}  // namespace

inline bool wide(const Frame& f) { return f.n > 1024; }           // plain ids are 8 bytes wide (Id64: the direct kernel of VP_ALGO_NAIVE)

// One row of "none" per id format for out-of-grid reads (jfa_seed.hip)
int ensure_none_rows(vp_ctx* ctx);
const void* none_row_id9(vp_ctx* ctx);
const void* none_row_id10(vp_ctx* ctx);
const void* none_row_idc(vp_ctx* ctx);           // word row (2048 x 4), its byte row (2048 x 1) right behind it

}  // namespace vp
