// jfa.cpp -- JFA::Compute back ends.  Host(): this library's CPU path with the semantics of
// /root/reference/vplib/src/jfa/sequential.cpp:7-127 (and jfa/openmp.cpp when parallel): Jacobi passes
// k = n/2 .. 1, 26-neighbour scan z,y,x, strict '<', copysign keeps the caller's sign.  Device(): the
// C ABI stages with the reference's timer labels.  Build with -ffp-contract=off.
#include "jfa/jfa.h"

#include <cmath>
#include <memory>
#include <string>
#include <utility>

#include "debug_utils.h"
#include "profiling.h"
#include "vp_runtime.h"

namespace JFA::detail {

namespace {

inline bool Bit(const uint32_t* w, size_t n, int x, int y, int z)
{
    const size_t i = static_cast<size_t>(x) + (static_cast<size_t>(y) + static_cast<size_t>(z) * n) * n;
    return (w[i >> 5] >> (i & 31)) & 1u;
}

}  // namespace

void Host(bool parallel, const uint32_t* words, size_t n, float vs, const float origin[3], float* sdf)
{
    const std::string L = parallel ? "OpenmpJFA" : "SequentialJFA";
    PROFILING_SCOPE(L);
    const int N = static_cast<int>(n);
    const size_t total = n * n * n;
    const float ox = origin[0], oy = origin[1], oz = origin[2];

    std::unique_ptr<Position[]> pos, posNext;
    std::unique_ptr<float[]> sdfNext;
    {
        PROFILING_SCOPE(L + "::Memory");
        pos = std::make_unique<Position[]>(total);
        posNext = std::make_unique<Position[]>(total);
        sdfNext = std::make_unique<float[]>(total);
    }
    {
        PROFILING_SCOPE(L + "::Initialization");
#pragma omp parallel for if (parallel) collapse(2) schedule(static)
        for (int z = 0; z < N; ++z)
            for (int y = 0; y < N; ++y)
                for (int x = 0; x < N; ++x) {
                    if (!Bit(words, n, x, y, z)) continue;                // unset voxels keep the caller's fill
                    bool border = false;
                    for (int dz = -1; dz <= 1 && !border; ++dz)
                        for (int dy = -1; dy <= 1 && !border; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                const int nx = x + dx, ny = y + dy, nz = z + dz;
                                if (nx < 0 || nx >= N || ny < 0 || ny >= N || nz < 0 || nz >= N || !Bit(words, n, nx, ny, nz)) {
                                    border = true;
                                    break;
                                }
                            }
                    const size_t i = static_cast<size_t>(x) + (static_cast<size_t>(y) + static_cast<size_t>(z) * n) * n;
                    if (border) {
                        sdf[i] = 0.0f;
                        pos[i] = Position(ox + (x * vs), oy + (y * vs), oz + (z * vs));
                    } else {
                        sdf[i] = INFINITY;
                    }
                }
    }
    {
        PROFILING_SCOPE(L + "::Processing");
        float* sIn = sdf;
        float* sOut = sdfNext.get();
        Position* pIn = pos.get();
        Position* pOut = posNext.get();
        for (int k = N / 2; k >= 1; k /= 2) {
#pragma omp parallel for if (parallel) collapse(2) schedule(static)
            for (int z = 0; z < N; ++z)
                for (int y = 0; y < N; ++y)
                    for (int x = 0; x < N; ++x) {
                        const size_t i = static_cast<size_t>(x) + (static_cast<size_t>(y) + static_cast<size_t>(z) * n) * n;
                        const Position here(ox + (x * vs), oy + (y * vs), oz + (z * vs));
                        float best = sIn[i];
                        Position bestPos = pIn[i];
                        for (int dz = -1; dz <= 1; ++dz) {
                            const int nz = z + dz * k;
                            if (nz < 0 || nz >= N) continue;
                            for (int dy = -1; dy <= 1; ++dy) {
                                const int ny = y + dy * k;
                                if (ny < 0 || ny >= N) continue;
                                for (int dx = -1; dx <= 1; ++dx) {
                                    const int nx = x + dx * k;
                                    if ((dx | dy | dz) == 0 || nx < 0 || nx >= N) continue;
                                    const size_t j = static_cast<size_t>(nx) + (static_cast<size_t>(ny) + static_cast<size_t>(nz) * n) * n;
                                    if (!(std::fabs(sIn[j]) < INFINITY)) continue;
                                    const float d = CalculateDistance(here, pIn[j]);
                                    if (d < std::fabs(best)) {
                                        best = std::copysign(d, best);
                                        bestPos = pIn[j];
                                    }
                                }
                            }
                        }
                        sOut[i] = best;
                        pOut[i] = bestPos;
                    }
            std::swap(sIn, sOut);
            std::swap(pIn, pOut);
        }
        if (sIn != sdf) std::copy_n(sIn, total, sdf);
    }
}

void Device(int algo, const char* label, const uint32_t* words, size_t n, float vs, const float origin[3], float* sdf)
{
    // Same launch sequence as vp_jfa (what the benchmark times): border mask, first pass straight from it, sparse / dense
    // tile passes, last pass fused with the id -> sdf conversion -- split at the point where the reference splits its
    // timers (jfa/tiled.cu:265-334).  Device buffers are the context's cached workspace: no allocation in steady state.
    const std::string L(label);
    PROFILING_SCOPE(L);
    vp_frame f{};
    f.n = static_cast<uint32_t>(n); f.voxel_size = vs;
    f.origin[0] = origin[0]; f.origin[1] = origin[1]; f.origin[2] = origin[2];
    f.z0 = 0; f.z1 = f.n;
    const size_t gridBytes = vp_grid_words(&f) * 4, voxels = vp_grid_voxels(&f);
    // the sign of unset voxels comes from the caller's pre-fill (apps/cli/main.cpp:200: -INFINITY)
    float fill = -INFINITY;
    for (size_t i = 0; i < voxels; ++i)
        if (!((words[i >> 5] >> (i & 31)) & 1u)) { fill = sdf[i]; break; }
    if (vp_multi* multi = vplib::Multi()) {                         // several devices: Z-slabs, halo copies or ghost planes
        {
            PROFILING_SCOPE(L + "::Memory");
            gpuAssert(vp_multi_set_grid(multi, &f, words));
        }
#if PROFILING
        vplib::MultiProfile(true);
#endif
        {
            // seeding and passes are ONE enqueue on every device (no host synchronisation inside vp_multi_jfa), so the reference's
            // ::Initialization / ::Processing split (jfa/tiled.cu:265-334) has no wall-clock boundary here: everything is
            // ::Processing, and the device-time lines below give the per-kernel split (surface / jfa_init = the seeding).
            PROFILING_SCOPE(L + "::Processing");
            gpuAssert(vp_multi_jfa(multi, fill, algo, vplib::MultiMode()));
            gpuAssert(vp_multi_sync(multi));
        }
#if PROFILING
        vplib::MultiProfile(false);
        vplib::PrintMultiDeviceTimes(L);
#endif
        {
            PROFILING_SCOPE(L + "::Memory");
            gpuAssert(vp_multi_get_sdf(multi, sdf));
        }
        return;
    }
    vp_ctx* ctx = vplib::Context();
    void *dWords = nullptr, *dSdf = nullptr;
    {
        PROFILING_SCOPE(L + "::Memory");
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotGridA, gridBytes, &dWords));
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotSdf, voxels * sizeof(float), &dSdf));
        gpuAssert(vp_upload(ctx, dWords, words, gridBytes));
    }
#if PROFILING
    gpuAssert(vp_prof_reset(ctx));
    gpuAssert(vp_prof_enable(ctx, 1));
#endif
    {
        PROFILING_SCOPE(L + "::Initialization");
        gpuAssert(vp_jfa_start(ctx, &f, static_cast<const uint32_t*>(dWords), nullptr, 0, algo));
        gpuAssert(vp_ctx_sync(ctx));
    }
    {
        PROFILING_SCOPE(L + "::Processing");
        gpuAssert(vp_jfa_run(ctx, &f, static_cast<const uint32_t*>(dWords), fill, static_cast<float*>(dSdf), nullptr, 0, algo));
        gpuAssert(vp_ctx_sync(ctx));
    }
#if PROFILING
    gpuAssert(vp_prof_enable(ctx, 0));
    vplib::PrintDeviceTimes(L);
#endif
    {
        PROFILING_SCOPE(L + "::Memory");
        gpuAssert(vp_download(ctx, sdf, dSdf, voxels * sizeof(float)));
    }
}

}  // namespace JFA::detail
