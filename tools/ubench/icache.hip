// Dev probe (round 4): does the LENGTH of a straight-line instruction stream cost issue rate on gfx950?
// The dense JFA tile kernel is one basic block of 35 - 65 KB per x iteration (the plane and row loops are fully unrolled so that
// every running (distance, rank) pair has a fixed register); the instruction cache is shared by CUs and a wave streams through that
// block once per 256 voxel columns.  This probe issues the SAME number of candidate steps (v_add_f32 into the high half of a pair +
// v_min_f64, the mix of the kernel: 4-byte and 8-byte encodings) from bodies of 1.5 KB ... 192 KB, every workgroup running the
// body `total / U` times, at 4 and 6 waves per SIMD.  If the time per step grows with the body, instruction fetch is a limiter of
// the tile kernel and its code should be made compact; if not, it is not.
// hipcc --offload-arch=gfx950 -O3 -o icache icache.hip && ./icache
#include <hip/hip_runtime.h>
#include <cstdio>

// one group = 4 candidate steps = 8 instructions = 48 bytes of code
#define GROUP asm volatile("v_add_f32 v101, %4, %5\n v_min_f64 %0, %0, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %1, %1, v[102:103]\n" \
                           "v_add_f32 v101, %4, %5\n v_min_f64 %2, %2, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %3, %3, v[102:103]"    \
                           : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "v100", "v101", "v102", "v103");

template <int U>
__global__ void __launch_bounds__(256) body(float* out, int reps, float seed)
{
    float a0 = seed + threadIdx.x, b = seed * 0.5f;
    double d0 = a0, d1 = a0 + 1, d2 = a0 + 2, d3 = a0 + 3;
    for (int r = 0; r < reps; ++r) {
#pragma clang loop unroll(full)
        for (int u = 0; u < U; ++u) { GROUP }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3);
}

template <int U>
static void run(float* out, int wgPerCu)
{
    const int totalGroups = 1 << 16, cus = 256;                 // 262,144 candidate steps per wave
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(body<U>, dim3(cus * wgPerCu), dim3(256), 0, 0, out, 2, 1.0f);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(body<U>, dim3(cus * wgPerCu), dim3(256), 0, 0, out, totalGroups / U, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double steps = (double)totalGroups * 4 * wgPerCu;     // per SIMD
    printf("body %7.1f KB  waves/SIMD=%d  %8.3f ms  %.2f clk@2.4GHz per candidate step per SIMD\n", U * 48 / 1024.0, wgPerCu, best, best * 1e6 / steps * 2.4);
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {4, 6}) {
        run<32>(out, w); run<128>(out, w); run<256>(out, w); run<512>(out, w); run<768>(out, w); run<1024>(out, w);
        run<1280>(out, w); run<1536>(out, w); run<2048>(out, w); run<4096>(out, w);
    }
    return 0;
}
