// Dev probe (round 4): which shape of a plain device copy reaches the box's HBM copy rate?  (vp_stream_copy, the yardstick bench.py
// reports as roofline.measured_copy_GBs, takes the best one.)  16 bytes per lane; U independent loads in flight per thread before the
// first store; B workgroups of 256 threads per CU; default / nt cache policy.  Rate = (bytes read + bytes written) / time, 1 GiB.
// hipcc --offload-arch=gfx950 -O3 -o copy copy.hip && ./copy
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_k(u4* __restrict__ dst, const u4* __restrict__ src, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < nvec; i += stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < nvec) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < nvec) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; }
        }
    }
}

template <int U, bool NT>
static void run(u4* dst, const u4* src, size_t bytes, int perCu)
{
    const size_t nvec = bytes / 16;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned blocks = 256u * perCu;
    hipLaunchKernelGGL((copy_k<U, NT>), dim3(blocks), dim3(256), 0, 0, dst, src, nvec);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((copy_k<U, NT>), dim3(blocks), dim3(256), 0, 0, dst, src, nvec);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("U=%d nt=%d wg/CU=%2d  %.4f ms  %7.1f GB/s\n", U, (int)NT, perCu, best, 2.0 * bytes / (best * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = 1ull << 30;
    u4 *a, *b;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 1, bytes);
    for (int perCu : {4, 8, 16, 32}) {
        run<1, false>(b, a, bytes, perCu); run<2, false>(b, a, bytes, perCu); run<4, false>(b, a, bytes, perCu); run<8, false>(b, a, bytes, perCu);
        run<1, true>(b, a, bytes, perCu); run<4, true>(b, a, bytes, perCu); run<8, true>(b, a, bytes, perCu);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    (void)hipEventRecord(e0); (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyAsync D2D          %.4f ms  %7.1f GB/s\n", ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
    return 0;
}
