// jfa.h -- JFA::Compute (/root/reference/vplib/src/jfa/jfa.h:19-20,42-43).
// `sdf` must be pre-filled by the caller (the CLI uses -INFINITY, apps/cli/main.cpp:200) and receives
// the signed SQUARED distance: 0 on inside-border voxels, > 0 inside, < 0 outside.
#ifndef VPLIB_JFA_H
#define VPLIB_JFA_H

#include "grid/grid.h"
#include "grid/voxels_grid.h"
#include "mesh/mesh.h"
#include "proc_utils.h"
#include "vphip.h"

namespace JFA {

inline float CalculateDistance(Position p0, Position p1)
{ return ((p1.X - p0.X) * (p1.X - p0.X)) + ((p1.Y - p0.Y) * (p1.Y - p0.Y)) + ((p1.Z - p0.Z) * (p1.Z - p0.Z)); }

namespace detail {
void Host(bool parallel, const uint32_t* words, size_t n, float voxelSize, const float origin[3], float* sdf);
void Device(int algo, const char* label, const uint32_t* words, size_t n, float voxelSize, const float origin[3], float* sdf);
}  // namespace detail

template <Types type, VGType T>
void Compute(HostVoxelsGrid<T>& grid, HostGrid<float>& sdf)
{
    auto& v = grid.View();
    const float origin[3] = {v.OriginX(), v.OriginY(), v.OriginZ()};
    const uint32_t* words = reinterpret_cast<const uint32_t*>(v.Data());
    float* out = sdf.View().Data();
    if constexpr (type == Types::SEQUENTIAL) detail::Host(false, words, v.VoxelsPerSide(), v.VoxelSize(), origin, out);
    else if constexpr (type == Types::OPENMP) detail::Host(true, words, v.VoxelsPerSide(), v.VoxelSize(), origin, out);
    else if constexpr (type == Types::NAIVE) detail::Device(VP_ALGO_NAIVE, "NaiveJFA", words, v.VoxelsPerSide(), v.VoxelSize(), origin, out);
    else detail::Device(VP_ALGO_TILED, "TiledJFA", words, v.VoxelsPerSide(), v.VoxelSize(), origin, out);
}

}  // namespace JFA

#endif
