// Dev microbenchmark: issue rate of a few VALU / LDS instructions on gfx950 (wave64 instructions per clock per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void __launch_bounds__(256) bench(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f;
    float half = (threadIdx.x & 1) ? 0.0f : 1.0f;          // half < b for every other lane
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = b;
    __shared__ float lds[1024];
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1; lds[threadIdx.x + 512] = a2; lds[threadIdx.x + 768] = a3;
    __syncthreads();
    unsigned addr = (threadIdx.x * 4u) & 4095u;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));) }
        if (OP == 2) { REP8(asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 3) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 4) { REP8(asm volatile("v_lshl_add_u64 %0, %8, 0, %0\n v_lshl_add_u64 %1, %8, 0, %1\n v_lshl_add_u64 %2, %8, 0, %2\n v_lshl_add_u64 %3, %8, 0, %3\n v_lshl_add_u64 %4, %8, 0, %4\n v_lshl_add_u64 %5, %8, 0, %5\n v_lshl_add_u64 %6, %8, 0, %6\n v_lshl_add_u64 %7, %8, 0, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));) }
        if (OP == 5) { REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));) }
        if (OP == 6) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 7) { REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 9) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %0, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %3, vcc\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 10) { REP8(asm volatile("s_mov_b64 s[20:21], exec\n v_cmpx_lt_f32 exec, %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %0\n s_mov_b64 exec, s[20:21]\n v_cmpx_lt_f32 exec, %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %3\n s_mov_b64 exec, s[20:21]\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "vcc");) }
        if (OP == 11) { REP8(asm volatile("v_cmp_lt_f32 vcc, %9, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %0, vcc\n v_cmp_lt_f32 vcc, %9, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %3, vcc\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(half) : "vcc");) }
        if (OP == 12) { REP8(asm volatile("s_mov_b64 s[20:21], exec\n v_cmpx_lt_f32 exec, %9, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %0\n s_mov_b64 exec, s[20:21]\n v_cmpx_lt_f32 exec, %9, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %3\n s_mov_b64 exec, s[20:21]\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(half) : "s20", "s21", "vcc");) }
        if (OP == 8) { REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int OP>
static void run(const char* name, float* out, int wgPerCu)
{
    const int iters = 2000, cus = 256;
    const int blocks = cus * wgPerCu;                     // 256-thread workgroups = 4 waves = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 64 * wgPerCu;   // wave-instructions per SIMD
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f clk @2.4GHz)\n", name, wgPerCu, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) { run<0>("v_add_f32", out, 1); run<1>("v_pk_add_f32", out, 1); run<2>("v_and_b32", out, 1); run<3>("v_cmp+v_cndmask (pairs)", out, 1); run<4>("v_lshl_add_u64", out, 1); run<5>("ds_read_b32 (linear)", out, 1); run<6>("v_cndmask_b32", out, 1); run<7>("v_cmp_lt_f32 -> sgpr", out, 1); run<8>("v_mul_f32", out, 1); }
        if (w == 2) { run<0>("v_add_f32", out, 2); run<1>("v_pk_add_f32", out, 2); run<2>("v_and_b32", out, 2); run<3>("v_cmp+v_cndmask (pairs)", out, 2); run<5>("ds_read_b32 (linear)", out, 2); }
        if (w == 4) { run<11>("half lanes: 2x(cmp+2 cndmask)+2 add", out, 4); run<12>("half lanes: 2x(cmpx+2 mov+s_mov)+2 add", out, 4); }
        if (w == 4) { run<9>("2x(cmp+2 cndmask)+2 add [8 instr]", out, 4); run<10>("2x(cmpx+2 mov+s_mov)+2 add [8 valu]", out, 4); }
        if (w == 4) { run<0>("v_add_f32", out, 4); run<1>("v_pk_add_f32", out, 4); run<2>("v_and_b32", out, 4); run<3>("v_cmp+v_cndmask (pairs)", out, 4); run<4>("v_lshl_add_u64", out, 4); run<5>("ds_read_b32 (linear)", out, 4); run<6>("v_cndmask_b32", out, 4); run<7>("v_cmp_lt_f32 -> sgpr", out, 4); }
        if (w == 8) { run<0>("v_add_f32", out, 8); run<5>("ds_read_b32 (linear)", out, 8); }
    }
    return 0;
}
