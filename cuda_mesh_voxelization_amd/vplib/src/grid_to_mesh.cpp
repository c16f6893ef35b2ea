// grid_to_mesh.cpp -- see grid_to_mesh.h.  Visual exports only; no parity contract beyond the OBJ conventions
// of the reference (six axis normals in the order +Z,+Y,+X,-Z,-Y,-X, grid_to_mesh.cpp:25-31; colours per vertex).
#include "mesh/grid_to_mesh.h"

#include <array>
#include <cstdint>
#include <unordered_map>

namespace {

void AxisNormals(Mesh& mesh)
{
    mesh.Normals = {Normal(0, 0, 1), Normal(0, 1, 0), Normal(1, 0, 0), Normal(0, 0, -1), Normal(0, -1, 0), Normal(-1, 0, 0)};
}

float Diagonal(float side) { return std::sqrt(side * side * 3.0f); }

}  // namespace

template <VGType T>
bool VoxelsGridToMeshCompressed(const VoxelsGrid<T>& grid, Mesh& mesh)
{
    mesh.Clear();
    AxisNormals(mesh);
    const int64_t n = static_cast<int64_t>(grid.VoxelsPerSide());
    const int64_t nv = n + 1;
    std::unordered_map<int64_t, uint32_t> vertexOf;               // lattice point -> vertex index (shared between faces)
    auto vertex = [&](int64_t x, int64_t y, int64_t z) -> uint32_t {
        const int64_t key = (z * nv + y) * nv + x;
        auto [it, fresh] = vertexOf.try_emplace(key, static_cast<uint32_t>(mesh.Coords.size()));
        if (fresh)
            mesh.Coords.emplace_back(grid.OriginX() + (x * grid.VoxelSize()), grid.OriginY() + (y * grid.VoxelSize()),
                                     grid.OriginZ() + (z * grid.VoxelSize()));
        return it->second;
    };
    auto set = [&](int64_t x, int64_t y, int64_t z) {
        return x >= 0 && y >= 0 && z >= 0 && x < n && y < n && z < n && grid.Voxel(x, y, z);
    };
    // the two in-plane axes of each face direction, ordered so that (u x v) points along +axis
    static const int U[3][3] = {{0, 1, 0}, {0, 0, 1}, {1, 0, 0}};   // axis X: u = Y ; axis Y: u = Z ; axis Z: u = X
    static const int V[3][3] = {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}};   // axis X: v = Z ; axis Y: v = X ; axis Z: v = Y
    static const uint32_t normalIndex[3][2] = {{5, 2}, {4, 1}, {3, 0}};   // [axis][positive side]
    for (int64_t z = 0; z < n; ++z)
        for (int64_t y = 0; y < n; ++y)
            for (int64_t x = 0; x < n; ++x) {
                if (!grid.Voxel(x, y, z)) continue;
                for (int axis = 0; axis < 3; ++axis)
                    for (int side = 0; side < 2; ++side) {
                        const int d = side ? 1 : -1;
                        const int64_t ax = axis == 0, ay = axis == 1, az = axis == 2;
                        if (set(x + d * ax, y + d * ay, z + d * az)) continue;       // interior face: not visible
                        const int64_t bx = x + side * ax, by = y + side * ay, bz = z + side * az;   // face corner
                        const uint32_t p00 = vertex(bx, by, bz);
                        const uint32_t p10 = vertex(bx + U[axis][0], by + U[axis][1], bz + U[axis][2]);
                        const uint32_t p01 = vertex(bx + V[axis][0], by + V[axis][1], bz + V[axis][2]);
                        const uint32_t p11 = vertex(bx + U[axis][0] + V[axis][0], by + U[axis][1] + V[axis][1], bz + U[axis][2] + V[axis][2]);
                        if (side) mesh.FacesCoords.insert(mesh.FacesCoords.end(), {p00, p10, p11, p00, p11, p01});   // outward = +axis
                        else      mesh.FacesCoords.insert(mesh.FacesCoords.end(), {p00, p11, p10, p00, p01, p11});   // outward = -axis
                        mesh.FacesNormals.insert(mesh.FacesNormals.end(), 6, normalIndex[axis][side]);
                    }
            }
    mesh.Colors.assign(mesh.VerticesSize(), Color(1.0f, 1.0f, 1.0f, 1.0f));
    return true;
}

template <VGType T>
bool VoxelsGridToMesh(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh)
{
    mesh.Clear();
    AxisNormals(mesh);
    const size_t n = grid.VoxelsPerSide();
    const float vs = grid.VoxelSize();
    const float max = Diagonal(n * vs);
    // corner c = dx + 2 dy + 4 dz ; two triangles per face, outward winding ; normal index per face
    static const uint32_t quads[6][4] = {{0, 2, 3, 1}, {4, 5, 7, 6}, {2, 6, 7, 3}, {0, 1, 5, 4}, {1, 3, 7, 5}, {0, 4, 6, 2}};
    static const uint32_t quadNormal[6] = {3, 0, 1, 4, 2, 5};     // -Z, +Z, +Y, -Y, +X, -X
    uint32_t cubes = 0;
    for (size_t z = 0; z < n; ++z)
        for (size_t y = 0; y < n; ++y)
            for (size_t x = 0; x < n; ++x) {
                if (!grid.Voxel(x, y, z) || std::isinf(sdf(x, y, z))) continue;
                const auto [r, g, b] = SDFToRGB(std::sqrt(sdf(x, y, z)), max);
                for (int c = 0; c < 8; ++c) {
                    mesh.Coords.emplace_back(grid.OriginX() + (x * vs) + (vs * (c & 1)), grid.OriginY() + (y * vs) + (vs * ((c >> 1) & 1)),
                                             grid.OriginZ() + (z * vs) + (vs * ((c >> 2) & 1)));
                    mesh.Colors.emplace_back(r, g, b, 1.0f);
                }
                const uint32_t base = cubes * 8;
                for (int q = 0; q < 6; ++q) {
                    const uint32_t* p = quads[q];
                    mesh.FacesCoords.insert(mesh.FacesCoords.end(), {base + p[0], base + p[1], base + p[2], base + p[0], base + p[2], base + p[3]});
                    mesh.FacesNormals.insert(mesh.FacesNormals.end(), 6, quadNormal[q]);
                }
                ++cubes;
            }
    mesh.ShrinkToFit();
    return true;
}

template <VGType T>
bool VoxelsGridToPointCloud(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh)
{
    mesh.Clear();
    const size_t n = grid.VoxelsPerSide();
    const float vs = grid.VoxelSize();
    const float max = Diagonal(n * vs);
    for (size_t z = 0; z < n; ++z)
        for (size_t y = 0; y < n; ++y)
            for (size_t x = 0; x < n; ++x) {
                if (!grid.Voxel(x, y, z)) continue;
                mesh.Coords.emplace_back(grid.OriginX() + (x * vs) + (vs / 2), grid.OriginY() + (y * vs) + (vs / 2),
                                         grid.OriginZ() + (z * vs) + (vs / 2));
                const auto [r, g, b] = SDFToRGB(std::sqrt(sdf(x, y, z)), max);
                mesh.Colors.emplace_back(r, g, b, 1.0f);
            }
    mesh.ShrinkToFit();
    return true;
}

template bool VoxelsGridToMeshCompressed<uint32_t>(const VoxelsGrid<uint32_t>&, Mesh&);
template bool VoxelsGridToMeshCompressed<uint64_t>(const VoxelsGrid<uint64_t>&, Mesh&);
template bool VoxelsGridToMesh<uint32_t>(const VoxelsGrid<uint32_t>&, const Grid<float>&, Mesh&);
template bool VoxelsGridToMesh<uint64_t>(const VoxelsGrid<uint64_t>&, const Grid<float>&, Mesh&);
template bool VoxelsGridToPointCloud<uint32_t>(const VoxelsGrid<uint32_t>&, const Grid<float>&, Mesh&);
template bool VoxelsGridToPointCloud<uint64_t>(const VoxelsGrid<uint64_t>&, const Grid<float>&, Mesh&);
