// csg.cpp -- CSG::Compute back ends (/root/reference/vplib/src/csg/sequential.cpp:7-30,
// csg/openmp.cpp:9-31, csg/naive.cu:26-64): word-wise a = a op b.
#include "csg/csg.h"

#include "profiling.h"
#include "vp_runtime.h"

namespace CSG::detail {

void Host(bool parallel, uint32_t* a, const uint32_t* b, size_t n, int op)
{
    const std::string L = parallel ? "OpenmpCSG" : "SequentialCSG";
    PROFILING_SCOPE(L);
    PROFILING_SCOPE(L + "::Processing");
    const long long count = static_cast<long long>(n);
    switch (static_cast<Op>(op)) {
        case Op::UNION:
#pragma omp parallel for if (parallel) schedule(static)
            for (long long i = 0; i < count; ++i) a[i] |= b[i];
            break;
        case Op::INTERSECTION:
#pragma omp parallel for if (parallel) schedule(static)
            for (long long i = 0; i < count; ++i) a[i] &= b[i];
            break;
        case Op::DIFFERENCE:
#pragma omp parallel for if (parallel) schedule(static)
            for (long long i = 0; i < count; ++i) a[i] &= ~b[i];
            break;
        case Op::VOID:
            break;
    }
}

void Device(uint32_t* a, const uint32_t* b, size_t n, int op)
{
    PROFILING_SCOPE("NaiveCSG");
    if (vp_multi* multi = vplib::Multi()) {                         // several devices: word-wise on the Z-slabs
        vp_frame f{};
        f.n = 32;
        while (static_cast<size_t>(f.n) * f.n * f.n / 32 < n) f.n += 32;   // the operation needs the side only to cut the slabs
        f.voxel_size = 1.0f; f.z0 = 0; f.z1 = f.n;
        cpuAssert(static_cast<size_t>(f.n) * f.n * f.n / 32 == n, "CSG on several devices needs a cubic grid with a side that is a multiple of 32");
        {
            PROFILING_SCOPE("NaiveCSG::Memory");
            gpuAssert(vp_multi_set_grid(multi, &f, a));
        }
        {
            PROFILING_SCOPE("NaiveCSG::Processing");
            gpuAssert(vp_multi_csg(multi, b, n, op));                 // uploads the slabs of b, combines, synchronises
        }
        {
            PROFILING_SCOPE("NaiveCSG::Memory");
            gpuAssert(vp_multi_get_grid(multi, a));
        }
        return;
    }
    vp_ctx* ctx = vplib::Context();
    void *da = nullptr, *db = nullptr;
    {
        PROFILING_SCOPE("NaiveCSG::Memory");
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotGridA, n * 4, &da));     // cached by the context
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotGridB, n * 4, &db));
        gpuAssert(vp_upload(ctx, da, a, n * 4));
        gpuAssert(vp_upload(ctx, db, b, n * 4));
    }
    {
        PROFILING_SCOPE("NaiveCSG::Processing");
        gpuAssert(vp_csg(ctx, static_cast<uint32_t*>(da), static_cast<const uint32_t*>(db), n, op));
        gpuAssert(vp_ctx_sync(ctx));
    }
    {
        PROFILING_SCOPE("NaiveCSG::Memory");
        gpuAssert(vp_download(ctx, a, da, n * 4));
    }
}

}  // namespace CSG::detail
