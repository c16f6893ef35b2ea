// grid_to_mesh.h -- voxel grid -> OBJ-ready meshes for the CLI's -e exports
// (/root/reference/vplib/src/mesh/grid_to_mesh.h:15-22,133-157; apps/cli/main.cpp:118-124,192-197,220-230).
//   VoxelsGridToMeshCompressed  cube faces with shared vertices (white): the reference's mesh -- every face of every set
//                               voxel once, interior faces included, its vertex order, winding and normal indices
//                               (grid_to_mesh.h:25-92; pinned face by face by tests/test_export.py)
//   VoxelsGridToSurfaceMesh     (this build's addition, `vpcli --surface-only`) only the faces between a set voxel and an unset /
//                               outside neighbour: the visible surface without the interior quads
//   VoxelsGridToMesh            one 8-vertex cube per set voxel with a finite sdf, coloured by SDFToRGB(sqrt(sdf), diag)
//   VoxelsGridToPointCloud      one vertex at the centre of every set voxel, same colouring
// Never on the timed path (benchmark mode disables -e, main.cpp:57).  The *Device variants (used by the CLI for -t 1 / -t 2)
// leave the O(n^3) walk over the grid to the GPU (vp_extract, include/vphip.h) and write byte-identical files.
#ifndef VPLIB_GRID_TO_MESH_H
#define VPLIB_GRID_TO_MESH_H

#include <algorithm>
#include <cmath>
#include <tuple>

#include "grid/grid.h"
#include "grid/voxels_grid.h"
#include "mesh/mesh.h"

// grid_to_mesh.h:15-22: blue (near) -> red (far), cube-root ramp
inline std::tuple<float, float, float> SDFToRGB(float v, float max)
{
    float t = std::max(0.0f, std::min(v, max)) / max;
    t = std::cbrt(t);
    return {t, 0.0f, 1.0f - t};
}

template <VGType T> bool VoxelsGridToMeshCompressed(const VoxelsGrid<T>& grid, Mesh& mesh);
template <VGType T> bool VoxelsGridToMesh(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh);
template <VGType T> bool VoxelsGridToPointCloud(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh);
template <VGType T> bool VoxelsGridToSurfaceMesh(const VoxelsGrid<T>& grid, Mesh& mesh);
template <VGType T> bool VoxelsGridToMeshCompressedDevice(const VoxelsGrid<T>& grid, Mesh& mesh);
template <VGType T> bool VoxelsGridToSurfaceMeshDevice(const VoxelsGrid<T>& grid, Mesh& mesh);
template <VGType T> bool VoxelsGridToMeshDevice(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh);
template <VGType T> bool VoxelsGridToPointCloudDevice(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh);

#endif
