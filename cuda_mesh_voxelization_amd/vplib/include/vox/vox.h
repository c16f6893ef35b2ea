// vox.h -- VOX::Compute, the voxelizer entry points of the reference
// (/root/reference/vplib/src/vox/vox.h:22-32,107-111).
//   Compute<Types::SEQUENTIAL | OPENMP>(grid, mesh)  CPU path, XOR-accumulates into `grid`
//                                                    (vox/sequential.cpp:6-63; the CLI runs SEQUENTIAL for -t 3 too)
//   Compute<Types::NAIVE>(grid, mesh)                GPU, one thread per triangle   (vox/naive.cu:86-122)
//   Compute<Types::TILED>(blockSize, grid, mesh)     GPU, tile-binned hybrid        (vox/tiled.cu:488-576)
// GPU variants replace the grid contents (as the reference's do, tiled.cu:572-575) and keep its
// origin / voxel size.  All variants produce the sequential path's bitmask.
#ifndef VPLIB_VOX_H
#define VPLIB_VOX_H

#include <cstddef>

#include "grid/voxels_grid.h"
#include "mesh/mesh.h"
#include "proc_utils.h"
#include "vphip.h"

namespace VOX {

inline float CalculateEdgeFunctionZY(const Position& V0, const Position& V1, float y, float z)
{ return ((z - V0.Z) * (V1.Y - V0.Y)) - ((y - V0.Y) * (V1.Z - V0.Z)); }

inline Normal CalculateNormalZY(const Position& V0, const Position& V1)
{ return Position(0, V1.Z - V0.Z, -(V1.Y - V0.Y)); }

inline Normal CalculateFaceNormal(const Position& V0, const Position& V1, const Position& V2)
{ return Vec3<float>::Cross(V1 - V0, V2 - V1); }

namespace detail {
// grid words viewed as uint32 (see voxels_grid.h on why this is layout-neutral)
void Sequential(uint32_t* words, size_t n, float voxelSize, const float origin[3], const Mesh& mesh);
void Device(int algo, const char* label, uint32_t* words, size_t n, float voxelSize, const float origin[3], const Mesh& mesh);
}  // namespace detail

template <Types type, VGType T>
void Compute(HostVoxelsGrid<T>& grid, const Mesh& mesh)
{
    auto& v = grid.View();
    const float origin[3] = {v.OriginX(), v.OriginY(), v.OriginZ()};
    uint32_t* words = reinterpret_cast<uint32_t*>(v.Data());
    if constexpr (type == Types::SEQUENTIAL || type == Types::OPENMP)
        detail::Sequential(words, v.VoxelsPerSide(), v.VoxelSize(), origin, mesh);
    else if constexpr (type == Types::NAIVE)
        detail::Device(VP_ALGO_NAIVE, "NaiveVox", words, v.VoxelsPerSide(), v.VoxelSize(), origin, mesh);
    else
        detail::Device(VP_ALGO_TILED, "TiledVox", words, v.VoxelsPerSide(), v.VoxelSize(), origin, mesh);
}

// blockSize is the reference's -b knob (threads per tile workgroup, tiled.cu:557-566); the HIP tile
// kernel has one fixed wave64-shaped workgroup, so the value is accepted and ignored.
template <Types type, VGType T>
void Compute(const size_t /*blockSize*/, HostVoxelsGrid<T>& grid, const Mesh& mesh)
{
    Compute<type, T>(grid, mesh);
}

}  // namespace VOX

#endif
