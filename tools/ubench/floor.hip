// Dev probe (round 4): the FLOOR of the exact 27-candidate scatter formulation of a JFA pass on gfx950.
//
// The tile kernel (csrc/jfa_dense.hip: jfa_pass_dense) evaluates, per voxel and pass, 27 candidates in float32 with a (distance, rank)
// minimum each (v_add_f32 into the high half of a pair + v_min_f64), after decoding the ids they come from.  This probe keeps
// nothing but that irreducible work and the stream it needs:
//
//   * an id volume of n^3 dwords is read ONCE and an output volume written ONCE, coalesced (one dword per lane and access, as in
//     the tile kernel), no halo rows or planes, no winner gather, no LDS tables, no table prologue, no tile index arithmetic;
//   * per voxel: 3 "ids" (the loaded one and two bit rotations of it: 2 instructions that the real kernel does not have, in
//     place of its two further loads), each decoded with 6 integer instructions (the IdU<10> decode of n = 1024; 4 with BITS9)
//     into seed x and the squared y / z differences the real kernel fetches from LDS (here: bit patterns of the id forced into
//     the normal float range -- live data, nothing the compiler can fold), then sub + mul (dx^2), 3 adds (dx^2 + dy^2 per output
//     row), 9 adds (+ dz^2 per output plane) and 9 v_min_f64 into nine running pairs, with one rank per id (1 instruction);
//   * per voxel one of the nine pairs is stored (its low word) and reset, so the stores are a coalesced stream too.
//
//   => 3 x (6 + 1 + 2 + 3 + 9 + 9) + 2 + ~3 = 95 vector instructions per voxel, 27 of them v_min_f64.
//
// Variant DIST: distances only (the fused last pass): v_min3_f32 pairs instead of the pair minimum, no ranks; one float stored.
// The time of FLOOR is a lower bound for ANY exact evaluation of this formulation on this chip at the occupancy given
// (waves per SIMD: 4 = the closed 8 x 8 tiles, 6 = the 4 x 8 tiles); bench.py quotes it as `formulation_floor_frac`.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o floor floor.hip && ./floor [n]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double min_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min3_f32(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

struct Dec { float sx, dy2[3], dz2[3]; uint32_t rank; };

// 6 integer instructions (BITS9: 4) turning an id into seven live floats in [2, 4) and a rank (1 instruction)
template <bool BITS9>
__device__ __forceinline__ Dec decode(uint32_t id, uint32_t rbase)
{
    Dec d;
    const uint32_t e = 0x40000000u;
    const uint32_t a = (id & 0x007FFFFFu) | e;                        // v_and_or_b32
    const uint32_t b = ((id << 7) & 0x007FFF80u) | e;                 // v_lshlrev + v_and_or   (2)
    const uint32_t c = ((id >> 9) & 0x007FFFFFu) | e;                 // v_lshrrev + v_and_or   (2)
    uint32_t g = c;
    if (!BITS9) g = (id & 0x00555555u) | e;                           // v_and_or_b32: the sixth (n = 1024 pays two more than n = 512)
    d.sx = __uint_as_float(a);
    d.dy2[0] = __uint_as_float(b); d.dy2[1] = __uint_as_float(c); d.dy2[2] = __uint_as_float(g);
    d.dz2[0] = __uint_as_float(c); d.dz2[1] = __uint_as_float(a); d.dz2[2] = __uint_as_float(b);
    d.rank = rbase + id;                                              // one v_add_u32: the candidate's low word, set once per id
    return d;
}

template <bool DIST, bool BITS9, int NT, int WAVES>
__global__ void __launch_bounds__(NT, WAVES)
floor_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t perBlock, float px, uint32_t rbase)
{
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * perBlock;
    const __amdgpu_buffer_rsrc_t rin = rsrc(in + base, perBlock * 4u), rout = rsrc(out + base, perBlock * 4u);
    double best[9];
    float bestf[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { best[i] = __builtin_bit_cast(double, (u32x2){0xFFFFFFFFu, 0x7F800000u}); bestf[i] = __builtin_inff(); }
    const uint32_t steps = perBlock / NT;                              // a multiple of 9
    uint32_t next = __builtin_amdgcn_raw_buffer_load_b32(rin, (int)(tid * 4u), 0, 0);
    for (uint32_t s0 = 0; s0 < steps; s0 += 9) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const uint32_t s = s0 + j;
            const uint32_t id0 = next;
            // the next voxel's id is requested a whole voxel of evaluation ahead (the tile kernel: a plane ahead)
            next = __builtin_amdgcn_raw_buffer_load_b32(rin, (int)(((s + 1 < steps ? s + 1 : s) * NT + tid) * 4u), 0, 0);
            float hold[9];
            const uint32_t ids[3] = {id0, __builtin_amdgcn_alignbit(id0, id0, 7), __builtin_amdgcn_alignbit(id0, id0, 13)};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const Dec d = decode<BITS9>(ids[c], rbase);
                const float dxv = d.sx - px;
                const float dx2 = dxv * dxv;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float pre = d.dy2[a] + dx2;
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float dd = d.dz2[o] + pre;
                        if (DIST) {
                            // as the fused last pass: the first column's distance waits for the second's, both go through one
                            // v_min3_f32, the third column takes a plain minimum: 18 minimum instructions per voxel
                            if (c == 0) hold[a * 3 + o] = dd;
                            else if (c == 1) bestf[a * 3 + o] = min3_f32(bestf[a * 3 + o], hold[a * 3 + o], dd);
                            else bestf[a * 3 + o] = fminf(bestf[a * 3 + o], dd);
                        } else {
                            u32x2 cd; cd.x = d.rank; cd.y = __float_as_uint(dd);
                            best[a * 3 + o] = min_f64(best[a * 3 + o], __builtin_bit_cast(double, cd));
                        }
                    }
                }
            }
            if (DIST) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(bestf[j]), rout, (int)((s * NT + tid) * 4u), 0, 0);
                bestf[j] = __builtin_inff();
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32x2, best[j]).x, rout, (int)((s * NT + tid) * 4u), 0, 0);
                best[j] = __builtin_bit_cast(double, (u32x2){0xFFFFFFFFu, 0x7F800000u});
            }
        }
    }
}

template <bool DIST, bool BITS9, int NT, int WAVES>
static double run(const char* name, const uint32_t* in, uint32_t* out, size_t voxels, double bytes)
{
    const uint32_t perBlock = NT * 9 * 8;                              // 72 voxels per thread
    const unsigned blocks = (unsigned)(voxels / perBlock);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((floor_kernel<DIST, BITS9, NT, WAVES>), dim3(blocks), dim3(NT), 0, 0, in, out, perBlock, 2.5f, 12345u);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((floor_kernel<DIST, BITS9, NT, WAVES>), dim3(blocks), dim3(NT), 0, 0, in, out, perBlock, 2.5f, 12345u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double covered = (double)blocks * perBlock / (double)voxels;
    const double gbs = bytes * covered / (best * 1e-3) / 1e9;
    printf("%-64s %8.3f ms  %7.1f GB/s algorithmic  frac_of_8TB/s %.3f  (%.1f ps per voxel)\n", name, best, gbs, gbs / 8000.0,
           best * 1e9 / ((double)blocks * perBlock));
    return gbs / 8000.0;
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atoi(argv[1]) : 1024;
    const size_t voxels = n * n * n;
    uint32_t *in, *out;
    if (hipMalloc(&in, voxels * 4) != hipSuccess || hipMalloc(&out, voxels * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    // ids with every bit live (a multiplicative hash of the index), so that the decodes see what a real volume gives them
    {
        uint32_t* h = (uint32_t*)malloc(voxels * 4);
        for (size_t i = 0; i < voxels; ++i) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
        (void)hipMemcpy(in, h, voxels * 4, hipMemcpyHostToDevice);
        free(h);
    }
    const double pass = 2.0 * 4.0 * (double)voxels;                    // 2 S n^3, S = 4
    printf("n = %zu: one id volume in, one out (%.2f GB algorithmic), irreducible work of the exact 27-candidate pass per voxel\n", n, pass / 1e9);
    const bool b9 = n <= 512;
    double f4, f6, d4, d6;
    if (b9) {
        f4 = run<false, true, 256, 4>("FLOOR  27 x (add + v_min_f64), 3 decodes, 4 waves/SIMD", in, out, voxels, pass);
        f6 = run<false, true, 256, 6>("FLOOR  27 x (add + v_min_f64), 3 decodes, 6 waves/SIMD", in, out, voxels, pass);
        d4 = run<true, true, 256, 4>("DIST   distances only, v_min3_f32 pairs, 4 waves/SIMD", in, out, voxels, pass);
        d6 = run<true, true, 256, 6>("DIST   distances only, v_min3_f32 pairs, 6 waves/SIMD", in, out, voxels, pass);
    } else {
        f4 = run<false, false, 512, 4>("FLOOR  27 x (add + v_min_f64), 3 decodes, 4 waves/SIMD", in, out, voxels, pass);
        f6 = run<false, false, 512, 6>("FLOOR  27 x (add + v_min_f64), 3 decodes, 6 waves/SIMD", in, out, voxels, pass);
        d4 = run<true, false, 512, 4>("DIST   distances only, v_min3_f32 pairs, 4 waves/SIMD", in, out, voxels, pass);
        d6 = run<true, false, 512, 6>("DIST   distances only, v_min3_f32 pairs, 6 waves/SIMD", in, out, voxels, pass);
    }
    printf("{\"n\": %zu, \"formulation_floor_frac\": %.4f, \"floor_frac_4_waves\": %.4f, \"floor_frac_6_waves\": %.4f, "
           "\"distance_only_floor_frac\": %.4f}\n", n, f4 > f6 ? f4 : f6, f4, f6, d4 > d6 ? d4 : d6);
    return 0;
}
