// vox.cpp -- VOX::Compute back ends: the CPU sequential path (this library's own Types::SEQUENTIAL,
// same semantics as /root/reference/vplib/src/vox/sequential.cpp:6-63) and the marshalling of the
// GPU variants onto the C ABI.  Build with -ffp-contract=off: a fused multiply-add changes bits.
#include "vox/vox.h"

#include <cmath>
#include <cstdio>

#include "debug_utils.h"
#include "profiling.h"
#include "vp_runtime.h"

namespace VOX::detail {

namespace {

// toggles linear bits [b0, b1) of a little-endian bit array
void FlipBits(uint32_t* words, size_t b0, size_t b1)
{
    if (b0 >= b1) return;
    const size_t w0 = b0 >> 5, w1 = (b1 - 1) >> 5;
    const uint32_t head = ~0u << (b0 & 31), tail = ~0u >> (31 - ((b1 - 1) & 31));
    if (w0 == w1) { words[w0] ^= head & tail; return; }
    words[w0] ^= head;
    for (size_t w = w0 + 1; w < w1; ++w) words[w] = ~words[w];
    words[w1] ^= tail;
}

}  // namespace

void Sequential(uint32_t* words, size_t n, float vs, const float origin[3], const Mesh& mesh)
{
    PROFILING_SCOPE("SequentialVox(" + mesh.Name + ")");
    PROFILING_SCOPE("SequentialVox::Processing");
    const float ox = origin[0], oy = origin[1], oz = origin[2];
    const int N = static_cast<int>(n);
    const size_t numTriangle = mesh.TrianglesSize();
    for (size_t t = 0; t < numTriangle; ++t) {
        const uint32_t* idx = &mesh.FacesCoords[3 * t];
        if (idx[0] >= mesh.Coords.size() || idx[1] >= mesh.Coords.size() || idx[2] >= mesh.Coords.size()) continue;
        const Position V0 = mesh.Coords[idx[0]], V1 = mesh.Coords[idx[1]], V2 = mesh.Coords[idx[2]];

        const float sign = CalculateFaceNormal(V0, V1, V2).X >= 0 ? 1.0f : -1.0f;

        float minY = V0.Y, maxY = V0.Y, minZ = V0.Z, maxZ = V0.Z;
        for (const Position* p : {&V1, &V2}) {
            if (p->Y < minY) minY = p->Y; else if (p->Y > maxY) maxY = p->Y;
            if (p->Z < minZ) minZ = p->Z; else if (p->Z > maxZ) maxZ = p->Z;
        }
        // columns outside the grid are skipped (the reference writes out of bounds there)
        const int startY = std::max(0, static_cast<int>(std::floor((minY - oy) / vs)));
        const int endY   = std::min(N, static_cast<int>(std::ceil((maxY - oy) / vs)));
        const int startZ = std::max(0, static_cast<int>(std::floor((minZ - oz) / vs)));
        const int endZ   = std::min(N, static_cast<int>(std::ceil((maxZ - oz) / vs)));

        const Position plane = Position::Cross(V1 - V0, V2 - V0);
        const float A = plane.X, B = plane.Y, C = plane.Z;
        const float D = Position::Dot(plane, V0);

        for (int y = startY; y < endY; ++y) {
            const float centerY = oy + ((y * vs) + (vs / 2));
            for (int z = startZ; z < endZ; ++z) {
                const float centerZ = oz + ((z * vs) + (vs / 2));
                const float E0 = CalculateEdgeFunctionZY(V0, V1, centerY, centerZ) * sign;
                const float E1 = CalculateEdgeFunctionZY(V1, V2, centerY, centerZ) * sign;
                const float E2 = CalculateEdgeFunctionZY(V2, V0, centerY, centerZ) * sign;
                if (!(E0 >= 0 && E1 >= 0 && E2 >= 0)) continue;
                const float intersection = (D - (B * centerY) - (C * centerZ)) / A;
                const float fx = (intersection - ox) / vs;
                if (!(fx > -2147483648.0f && fx < 2147483648.0f)) continue;      // A == 0: undefined in the reference
                const int startX = std::max(0, static_cast<int>(fx));
                if (startX >= N) continue;
                const size_t row = (static_cast<size_t>(z) * n + static_cast<size_t>(y)) * n;
                FlipBits(words, row + startX, row + n);
            }
        }
    }
}

void Device(int algo, const char* label, uint32_t* words, size_t n, float vs, const float origin[3], const Mesh& mesh)
{
    const std::string L(label);
    PROFILING_SCOPE(L + "(" + mesh.Name + ")");
    vp_frame f{};
    f.n = static_cast<uint32_t>(n); f.voxel_size = vs;
    f.origin[0] = origin[0]; f.origin[1] = origin[1]; f.origin[2] = origin[2];
    f.z0 = 0; f.z1 = f.n;
    const size_t nverts = mesh.Coords.size(), ntris = mesh.TrianglesSize();
    const size_t gridBytes = vp_grid_words(&f) * 4;
    if (vp_multi* multi = vplib::Multi()) {                         // several devices: every one rasterises its own Z-slab
        {
            PROFILING_SCOPE(L + "::Memory");
            gpuAssert(vp_multi_set_mesh(multi, reinterpret_cast<const float*>(mesh.Coords.data()), nverts, mesh.FacesCoords.data(), ntris));
        }
#if PROFILING
        vplib::MultiProfile(true);
#endif
        {
            PROFILING_SCOPE(L + "::Processing");
            gpuAssert(vp_multi_voxelize(multi, &f, algo));
            gpuAssert(vp_multi_sync(multi));
        }
#if PROFILING
        vplib::MultiProfile(false);
        if (algo == VP_ALGO_TILED) {                                // the TileAssignment columns of the benchmark CSV: slowest device
            std::printf("[TiledVox::TileAssignment::CalculateOverlap]: %f ms\n", vplib::MultiDeviceTime(VP_K_VOX_SETUP));
            std::printf("[TiledVox::TileAssignment::ExclusiveScan]: %f ms\n", vplib::MultiDeviceTime(VP_K_VOX_SCAN));
            std::printf("[TiledVox::TileAssignment::WorkQueuePopulation]: %f ms\n", vplib::MultiDeviceTime(VP_K_VOX_SCATTER));
            std::printf("[TiledVox::TileAssignment::WorkQueueSorting]: %f ms\n", 0.0);
            std::printf("[TiledVox::TileAssignment::CompactResult]: %f ms\n", 0.0);
        }
        vplib::PrintMultiDeviceTimes(L);
#endif
        {
            PROFILING_SCOPE(L + "::Memory");
            gpuAssert(vp_multi_get_grid(multi, words));
        }
        return;
    }
    vp_ctx* ctx = vplib::Context();
    void *dWords = nullptr, *dXyz = nullptr, *dTri = nullptr;
    {
        PROFILING_SCOPE(L + "::Memory");
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotGridA, gridBytes, &dWords));     // cached by the context: no allocation in steady state
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotXyz, nverts * sizeof(Position), &dXyz));
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotTri, ntris * 3 * sizeof(uint32_t), &dTri));
        gpuAssert(vp_upload(ctx, dXyz, mesh.Coords.data(), nverts * sizeof(Position)));
        gpuAssert(vp_upload(ctx, dTri, mesh.FacesCoords.data(), ntris * 3 * sizeof(uint32_t)));
    }
#if PROFILING
    gpuAssert(vp_prof_reset(ctx));
    gpuAssert(vp_prof_enable(ctx, 1));
#endif
    {
        PROFILING_SCOPE(L + "::Processing");
        gpuAssert(vp_voxelize(ctx, &f, static_cast<uint32_t*>(dWords), static_cast<const float*>(dXyz), nverts,
                              static_cast<const uint32_t*>(dTri), ntris, algo, /*accumulate=*/0));
        gpuAssert(vp_ctx_sync(ctx));
    }
#if PROFILING
    gpuAssert(vp_prof_enable(ctx, 0));
    if (algo == VP_ALGO_TILED) {
        // Device time of the binning stages under the reference's TileAssignment labels (vox/tiled.cu:31-236), so
        // that its benchmark CSV columns keep their meaning: setup = overlap test + histogram, scan = offsets,
        // scatter = work-queue population.  Nothing is sorted or compacted here (XOR accumulation is order-free).
        auto ms = [&](int kernel) { double t = 0; uint64_t c = 0; gpuAssert(vp_prof_get(ctx, kernel, &t, &c)); return t; };
        std::printf("[TiledVox::TileAssignment::CalculateOverlap]: %f ms\n", ms(VP_K_VOX_SETUP));
        std::printf("[TiledVox::TileAssignment::ExclusiveScan]: %f ms\n", ms(VP_K_VOX_SCAN));
        std::printf("[TiledVox::TileAssignment::WorkQueuePopulation]: %f ms\n", ms(VP_K_VOX_SCATTER));
        std::printf("[TiledVox::TileAssignment::WorkQueueSorting]: %f ms\n", 0.0);
        std::printf("[TiledVox::TileAssignment::CompactResult]: %f ms\n", 0.0);
    }
#endif
    {
        PROFILING_SCOPE(L + "::Memory");
        gpuAssert(vp_download(ctx, words, dWords, gridBytes));
    }
}

}  // namespace VOX::detail
