// csg.hip -- word-wise CSG of two bitmasks (reference: /root/reference/vplib/src/csg/csg.h:14-30,
// csg/sequential.cpp:7-30, csg/naive.cu:7-64).  Pure HBM streaming: 2 reads + 1 write of n^3/8
// bytes, 16 bytes per lane, grid-stride over a grid sized for 256 CUs.
#include "vp_internal.h"

namespace vp {

namespace {

template <int OP>
__device__ __forceinline__ uint32_t apply(uint32_t a, uint32_t b)
{
    if (OP == VP_OP_UNION) return a | b;            // csg.h:17
    if (OP == VP_OP_INTERSECTION) return a & b;     // csg.h:23
    return a & ~b;                                  // csg.h:29
}

template <int OP>
__global__ void __launch_bounds__(256)
csg_words(uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t nwords)
{
    const size_t nvec = nwords / 4;
    uint4* a4 = reinterpret_cast<uint4*>(a);
    const uint4* b4 = reinterpret_cast<const uint4*>(b);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint4 x = a4[i];
        const uint4 y = b4[i];
        x.x = apply<OP>(x.x, y.x); x.y = apply<OP>(x.y, y.y);
        x.z = apply<OP>(x.z, y.z); x.w = apply<OP>(x.w, y.w);
        a4[i] = x;
    }
    // tail (nwords not a multiple of 4)
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride)
        a[i] = apply<OP>(a[i], b[i]);
}

// The yardstick the HBM-bound kernels are priced against on the box they run on ("measure a device stream-copy
// peak on the box"): the same access shape as csg_words and vox_fill -- 16 bytes per lane, grid-stride -- with nothing computed.
// Shape (tools/ubench/copy.hip, profiles/r04/copy_shapes.txt): four independent loads in flight per thread before the first store,
// nt cache policy on both sides, 16 - 32 workgroups per CU: 6.06 TB/s on the round-4 box; one load in flight and the default policy
// (the round's first form): 4.87; hipMemcpyAsync device-to-device: 5.03.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kCopyUnroll = 4;
__global__ void __launch_bounds__(256)
stream_copy(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * 256 * kCopyUnroll;
    for (size_t i = (size_t)blockIdx.x * 256 * kCopyUnroll + threadIdx.x; i < nvec; i += stride) {
        u32x4 v[kCopyUnroll];
#pragma unroll
        for (int u = 0; u < kCopyUnroll; ++u) { const size_t j = i + (size_t)u * 256; if (j < nvec) v[u] = __builtin_nontemporal_load(src + j); }
#pragma unroll
        for (int u = 0; u < kCopyUnroll; ++u) { const size_t j = i + (size_t)u * 256; if (j < nvec) __builtin_nontemporal_store(v[u], dst + j); }
    }
}

}  // namespace

int launch_stream_copy(vp_ctx* ctx, void* d_dst, const void* d_src, size_t bytes)
{
    const size_t nvec = bytes / 16;
    const unsigned blocks = (unsigned)std::min<size_t>((nvec + 256 * kCopyUnroll - 1) / (256 * kCopyUnroll), (size_t)ctx->cus * 32);
    hipLaunchKernelGGL(stream_copy, dim3(blocks), dim3(256), 0, ctx->stream, (u32x4*)d_dst, (const u32x4*)d_src, nvec);
    VP_HIP(hipGetLastError());
    return 0;
}

int launch_csg(vp_ctx* ctx, uint32_t* d_a, const uint32_t* d_b, size_t nwords, int op)
{
    if (op == VP_OP_VOID || nwords == 0) return 0;
    const size_t nvec = (nwords + 3) / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((nvec + 255) / 256, 256 * 8);
    ProfScope p(ctx, VP_K_CSG);
    switch (op) {
        case VP_OP_UNION:
            hipLaunchKernelGGL(csg_words<VP_OP_UNION>, dim3(blocks), dim3(256), 0, ctx->stream, d_a, d_b, nwords);
            break;
        case VP_OP_INTERSECTION:
            hipLaunchKernelGGL(csg_words<VP_OP_INTERSECTION>, dim3(blocks), dim3(256), 0, ctx->stream, d_a, d_b, nwords);
            break;
        case VP_OP_DIFFERENCE:
            hipLaunchKernelGGL(csg_words<VP_OP_DIFFERENCE>, dim3(blocks), dim3(256), 0, ctx->stream, d_a, d_b, nwords);
            break;
        default:
            return set_error(VP_ERR_INVALID, "vp_csg: unknown op %d", op);
    }
    VP_HIP(hipGetLastError());
    return 0;
}

}  // namespace vp
