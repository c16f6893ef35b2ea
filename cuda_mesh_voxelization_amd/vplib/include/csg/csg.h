// csg.h -- CSG::Compute and its functors (/root/reference/vplib/src/csg/csg.h:10-36).
// result lands in grid1; both grids must have the same shape and voxel size (csg/naive.cu:30-33).
#ifndef VPLIB_CSG_H
#define VPLIB_CSG_H

#include <type_traits>

#include "debug_utils.h"
#include "grid/voxels_grid.h"
#include "proc_utils.h"

namespace CSG {

enum class Op { VOID, UNION, INTERSECTION, DIFFERENCE };

template <typename T> struct Union        { static constexpr Op kOp = Op::UNION;        void operator()(T& el, T v) const { el |= v; } };
template <typename T> struct Intersection { static constexpr Op kOp = Op::INTERSECTION; void operator()(T& el, T v) const { el &= v; } };
template <typename T> struct Difference   { static constexpr Op kOp = Op::DIFFERENCE;   void operator()(T& el, T v) const { el &= ~v; } };

namespace detail {
void Host(bool parallel, uint32_t* a, const uint32_t* b, size_t nwords32, int op);
void Device(uint32_t* a, const uint32_t* b, size_t nwords32, int op);
}  // namespace detail

template <Types type, VGType T, typename func>
void Compute(HostVoxelsGrid<T>& grid1, HostVoxelsGrid<T>& grid2, func)
{
    auto& a = grid1.View();
    auto& b = grid2.View();
    cpuAssert(a.SizeX() == b.SizeX() && a.SizeY() == b.SizeY() && a.SizeZ() == b.SizeZ(), "grid1 and grid2 must have same dimension");
    cpuAssert(a.VoxelSize() == b.VoxelSize(), "grid1 and grid2 must have same voxel size");
    const size_t n32 = a.StorageSize() * (sizeof(T) / 4);
    const int op = static_cast<int>(func::kOp);
    uint32_t* pa = reinterpret_cast<uint32_t*>(a.Data());
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(b.Data());
    if constexpr (type == Types::SEQUENTIAL) detail::Host(false, pa, pb, n32, op);
    else if constexpr (type == Types::OPENMP) detail::Host(true, pa, pb, n32, op);
    else detail::Device(pa, pb, n32, op);              // NAIVE and TILED share one kernel (apps/cli/main.cpp:167-185)
}

}  // namespace CSG

#endif
