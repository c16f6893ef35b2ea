// Dev probe (round 3): facts about gfx950 the pair-sharing JFA tile kernel relies on.
//   1. issue rate of VOP2 DPP forms (v_add_f32_dpp / v_sub_f32_dpp) next to the plain instructions, alone and inside the
//      candidate step (add -> high half of the pair, v_min_f64);
//   2. what quad_perm:[1,0,3,2], quad_perm:[2,3,0,1], row_half_mirror and row_ror:8 deliver, lane by lane;
//   3. the VALU-write -> DPP-read hazard: is the result right when the producer is the instruction right before
//      (inline asm is opaque to the compiler's hazard recogniser);
//   4. ds_read2_b32 / ds_read2st64_b32 against two ds_read_b32;
//   5. the hierarchical candidate update the round-2 review asked about: 3 x v_add_f32 + v_min3_f32 + 1 x v_min_f64
//      against 3 x (v_add_f32 + v_min_f64).
// hipcc --offload-arch=gfx950 -O3 -o probe3 probe3.hip && ./probe3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x

template <int OP>
__global__ void __launch_bounds__(256) rate(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b = seed * 0.5f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a0 + i;
    __syncthreads();
    unsigned addr4 = (threadIdx.x * 4u) & 4095u;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_add_f32_dpp %0, %4, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %4, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                                         "v_add_f32_dpp %2, %4, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                                         "v_add_f32_dpp %0, %4, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %4, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                                         "v_add_f32_dpp %2, %4, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 2) { REP8(asm volatile("v_add_f32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %4, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                         "v_add_f32_dpp %2, %4, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                         "v_add_f32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %4, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                         "v_add_f32_dpp %2, %4, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        // candidate step with a DPP source: add_dpp -> high half, min_f64   (4 steps)
        if (OP == 3) { REP8(asm volatile("v_add_f32_dpp v101, %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f64 %0, %0, v[100:101]\n v_add_f32_dpp v103, %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f64 %1, %1, v[102:103]\n"
                                         "v_add_f32_dpp v101, %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f64 %2, %2, v[100:101]\n v_add_f32_dpp v103, %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f64 %3, %3, v[102:103]"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "v100", "v101", "v102", "v103");) }
        if (OP == 4) { REP8(asm volatile("v_add_f32 v101, %4, %5\n v_min_f64 %0, %0, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %1, %1, v[102:103]\n"
                                         "v_add_f32 v101, %4, %5\n v_min_f64 %2, %2, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %3, %3, v[102:103]"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "v100", "v101", "v102", "v103");) }
        // hierarchical: 3 adds + min3 -> high half, ONE min_f64 per triple   (4 triples = 12 candidates)
        if (OP == 5) { REP8(asm volatile("v_add_f32 v104, %4, %5\n v_add_f32 v105, %4, %5\n v_add_f32 v106, %4, %5\n v_min3_f32 v101, v104, v105, v106\n v_min_f64 %0, %0, v[100:101]\n"
                                         "v_add_f32 v104, %4, %5\n v_add_f32 v105, %4, %5\n v_add_f32 v106, %4, %5\n v_min3_f32 v103, v104, v105, v106\n v_min_f64 %1, %1, v[102:103]\n"
                                         "v_add_f32 v104, %4, %5\n v_add_f32 v105, %4, %5\n v_add_f32 v106, %4, %5\n v_min3_f32 v101, v104, v105, v106\n v_min_f64 %2, %2, v[100:101]\n"
                                         "v_add_f32 v104, %4, %5\n v_add_f32 v105, %4, %5\n v_add_f32 v106, %4, %5\n v_min3_f32 v103, v104, v105, v106\n v_min_f64 %3, %3, v[102:103]"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "v100", "v101", "v102", "v103", "v104", "v105", "v106");) }
        // the same 12 candidates the round-2 way
        if (OP == 6) { REP8(asm volatile("v_add_f32 v101, %4, %5\n v_min_f64 %0, %0, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %0, %0, v[102:103]\n v_add_f32 v101, %4, %5\n v_min_f64 %0, %0, v[100:101]\n"
                                         "v_add_f32 v103, %4, %5\n v_min_f64 %1, %1, v[102:103]\n v_add_f32 v101, %4, %5\n v_min_f64 %1, %1, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %1, %1, v[102:103]\n"
                                         "v_add_f32 v101, %4, %5\n v_min_f64 %2, %2, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %2, %2, v[102:103]\n v_add_f32 v101, %4, %5\n v_min_f64 %2, %2, v[100:101]\n"
                                         "v_add_f32 v103, %4, %5\n v_min_f64 %3, %3, v[102:103]\n v_add_f32 v101, %4, %5\n v_min_f64 %3, %3, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %3, %3, v[102:103]"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "v100", "v101", "v102", "v103");) }
        if (OP == 7) { REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:2048\n ds_read_b32 %2, %4 offset:4096\n ds_read_b32 %3, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr4));) }
        // the same four dwords with two ds_read2st64_b32 (offsets in units of 256 bytes)
        if (OP == 8) { REP8(asm volatile("ds_read2st64_b32 %0, %2 offset0:0 offset1:8\n ds_read2st64_b32 %1, %2 offset0:16 offset1:24\n s_waitcnt lgkmcnt(0)"
                                         : "=v"(d0), "=v"(d1) : "v"(addr4));) }
        if (OP == 9) { REP8(asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:128\n ds_read2_b32 %1, %2 offset0:64 offset1:192\n s_waitcnt lgkmcnt(0)"
                                         : "=v"(d0), "=v"(d1) : "v"(addr4));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3);
}

template <int OP>
static void run(const char* name, float* out, int wgPerCu, double perIter)
{
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(cus * wgPerCu), dim3(256), 0, 0, out, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(cus * wgPerCu), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double units = (double)iters * 8 * perIter * wgPerCu;   // units per SIMD
    printf("%-58s waves/SIMD=%d  %8.3f ms  %.2f clk@2.4GHz per unit per SIMD\n", name, wgPerCu, ms, ms * 1e6 / units * 2.4);
}

// ---------------------------------------------------------------- 2. / 3. what the permutations deliver, hazard
__global__ void perms(uint32_t* out)
{
    const uint32_t lane = threadIdx.x;
    uint32_t v = 0x100u + lane, r0, r1, r2, r3;
    asm volatile("s_nop 4\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r0) : "v"(v));
    asm volatile("s_nop 4\n v_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r1) : "v"(v));
    asm volatile("s_nop 4\n v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r2) : "v"(v));
    asm volatile("s_nop 4\n v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r3) : "v"(v));
    out[lane] = r0; out[64 + lane] = r1; out[128 + lane] = r2; out[192 + lane] = r3;
    // hazard: producer right before the DPP consumer, no wait states in between; expected = partner's (lane * 3 + 7)
    uint32_t h0, h1, t;
    asm volatile("v_mad_u32_u24 %1, %2, 3, 7\n v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(h0), "=&v"(t) : "v"(lane));
    asm volatile("v_mad_u32_u24 %1, %2, 3, 7\n s_nop 1\n v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(h1), "=&v"(t) : "v"(lane));
    out[256 + lane] = h0; out[320 + lane] = h1;
    // the builtin form, to see whether the compiler folds it into the consumer (look at the ISA) and keeps the hazard right
    const float f = (float)lane * 1.5f;
    const float g = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x128, 0xf, 0xf, true)) + 100.0f;   // row_ror:8
    out[384 + lane] = __builtin_bit_cast(uint32_t, g);
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {4, 6, 8}) {
        run<0>("v_add_f32 (unit = 1 instr)", out, w, 8);
        run<1>("v_add_f32_dpp row_ror:8 (unit = 1 instr)", out, w, 8);
        run<2>("v_add_f32_dpp quad_perm (unit = 1 instr)", out, w, 8);
        run<4>("step: add_f32 + min_f64 (unit = step)", out, w, 4);
        run<3>("step: add_f32_dpp + min_f64 (unit = step)", out, w, 4);
        run<6>("12 candidates: 12 x (add + min_f64) (unit = candidate)", out, w, 12);
        run<5>("12 candidates: 4 x (3 add + min3 + min_f64) (unit = cand.)", out, w, 12);
        run<7>("4 x ds_read_b32 (unit = dword)", out, w, 4);
        run<8>("2 x ds_read2st64_b32 (unit = dword)", out, w, 4);
        run<9>("2 x ds_read2_b32 (unit = dword)", out, w, 4);
    }
    uint32_t* dout; (void)hipMalloc(&dout, 448 * 4);
    hipLaunchKernelGGL(perms, dim3(1), dim3(64), 0, 0, dout);
    std::vector<uint32_t> o(448); (void)hipMemcpy(o.data(), dout, 448 * 4, hipMemcpyDeviceToHost);
    auto show = [&](const char* name, int off, uint32_t sub) { printf("%-26s:", name); for (int l = 0; l < 32; ++l) printf(" %x", o[off + l] - sub); printf(" ...\n"); };
    show("quad_perm:[1,0,3,2]", 0, 0x100); show("quad_perm:[2,3,0,1]", 64, 0x100); show("row_half_mirror", 128, 0x100); show("row_ror:8", 192, 0x100);
    int bad0 = 0, bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t partner = (uint32_t)(l ^ 8);
        bad0 += o[256 + l] != partner * 3 + 7; bad1 += o[320 + l] != partner * 3 + 7;
        const float e = (float)partner * 1.5f + 100.0f;
        bad2 += o[384 + l] != __builtin_bit_cast(uint32_t, e);
    }
    printf("hazard: producer immediately before the DPP read: %d wrong lanes; with s_nop 1: %d wrong lanes; builtin update_dpp: %d wrong lanes\n", bad0, bad1, bad2);
    return 0;
}
