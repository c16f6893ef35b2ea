// grid_to_mesh.cpp -- see grid_to_mesh.h.  VoxelsGridToMeshCompressed emits the reference's mesh (grid_to_mesh.h:25-92,
// grid_to_mesh.cpp:10-60: face set, vertex order, winding, normal indices; pinned face by face by tests/test_export.py);
// VoxelsGridToMesh / VoxelsGridToPointCloud emit the reference's cubes and points (:65-201: vertex order, its twelve triangles and normal slots
// per cube, SDFToRGB colours through the 8-bit Color; pinned line by line by tests/test_export.py).
//
// Every exporter is written against a stream of voxel RECORDS (linear index + exposed-face mask) in z, y, x order.
// The host variants produce the records by walking the grid (what the reference does, grid_to_mesh.cpp:10-201); the
// *Device variants get them from the GPU compaction (vp_extract_count / vp_extract, include/vphip.h) and emit the same
// bytes -- the O(n^3) walk leaves the CPU, which then only touches the voxels it writes.
#include "mesh/grid_to_mesh.h"

#include <array>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "debug_utils.h"
#include "vp_runtime.h"

namespace {

void AxisNormals(Mesh& mesh)
{
    mesh.Normals = {Normal(0, 0, 1), Normal(0, 1, 0), Normal(1, 0, 0), Normal(0, 0, -1), Normal(0, -1, 0), Normal(-1, 0, 0)};
}

// grid_to_mesh.cpp:84,182: std::sqrt(std::pow(<float side>, 2) * 3) -- std::pow(float, int) and the sqrt of its result are DOUBLE operations,
// narrowed to float on assignment
float Diagonal(float side) { return static_cast<float>(std::sqrt(std::pow(static_cast<double>(side), 2.0) * 3.0)); }

constexpr uint64_t kIndexMask = (1ull << 40) - 1;

// host record generator, the same records vp_extract produces: mode VP_EXTRACT_SET = set voxels; VP_EXTRACT_EXPOSED = set voxels with a
// face towards unset / outside, + face mask (bit = axis * 2 + side; X, Y, Z; side 0 = minus); VP_EXTRACT_FACES = every set voxel + that mask
template <VGType T>
std::vector<uint64_t> HostRecords(const VoxelsGrid<T>& grid, int mode)
{
    const bool exposedOnly = mode == VP_EXTRACT_EXPOSED;
    std::vector<uint64_t> out;
    const int64_t n = static_cast<int64_t>(grid.VoxelsPerSide());
    auto set = [&](int64_t x, int64_t y, int64_t z) {
        return x >= 0 && y >= 0 && z >= 0 && x < n && y < n && z < n && grid.Voxel(x, y, z);
    };
    for (int64_t z = 0; z < n; ++z)
        for (int64_t y = 0; y < n; ++y)
            for (int64_t x = 0; x < n; ++x) {
                if (!grid.Voxel(x, y, z)) continue;
                const uint64_t idx = static_cast<uint64_t>(x + n * (y + n * z));
                if (mode == VP_EXTRACT_SET) { out.push_back(idx); continue; }
                uint64_t mask = 0;
                for (int axis = 0; axis < 3; ++axis)
                    for (int side = 0; side < 2; ++side) {
                        const int d = side ? 1 : -1;
                        if (!set(x + d * (axis == 0), y + d * (axis == 1), z + d * (axis == 2))) mask |= 1ull << (axis * 2 + side);
                    }
                if (mask || !exposedOnly) out.push_back(idx | (mask << 40));
            }
    return out;
}

// device record generator: upload the grid, count, extract, download
template <VGType T>
std::vector<uint64_t> DeviceRecords(const VoxelsGrid<T>& grid, int mode)
{
    vp_ctx* ctx = vplib::Context();
    vp_frame f{};
    f.n = static_cast<uint32_t>(grid.VoxelsPerSide()); f.voxel_size = grid.VoxelSize();
    f.origin[0] = grid.OriginX(); f.origin[1] = grid.OriginY(); f.origin[2] = grid.OriginZ();
    f.z0 = 0; f.z1 = f.n;
    const size_t gridBytes = vp_grid_words(&f) * 4;
    void *dWords = nullptr, *dRec = nullptr;
    gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotGridA, gridBytes, &dWords));
    gpuAssert(vp_upload(ctx, dWords, grid.Data(), gridBytes));
    uint64_t count = 0;
    gpuAssert(vp_extract_count(ctx, &f, static_cast<const uint32_t*>(dWords), mode, &count));
    std::vector<uint64_t> out(count);
    if (count) {
        gpuAssert(vp_ctx_workspace(ctx, vplib::kSlotRecords, count * sizeof(uint64_t), &dRec));
        gpuAssert(vp_extract(ctx, &f, static_cast<const uint32_t*>(dWords), mode, nullptr, static_cast<uint64_t*>(dRec), nullptr, count));
        gpuAssert(vp_download(ctx, out.data(), dRec, count * sizeof(uint64_t)));
    }
    return out;
}

// ---- emitters (shared by the host and the device variants) ------------------------------------------------
// The reference's compressed mesh (grid_to_mesh.cpp:10-60): for every set voxel in z, y, x order the faces XY back / front, XZ back / front,
// YZ back / front (:37-44), each emitted ONCE -- a face already emitted by the voxel on its other side is skipped (faces_marked,
// grid_to_mesh.h:36-41); in scan order that is exactly "a back face is skipped iff the voxel behind it is set" (that voxel came first and
// emitted it as its front face), which is what the face mask of the records says.  Vertices are shared through a map keyed by lattice
// point and numbered in order of first use, four per face in (v, u) order (:45-65); the two triangles and their winding depend on plane
// and side (:67-85); six normal indices (front * 3 + plane_index) per face (:87).
template <VGType T>
void EmitReferenceFaces(const VoxelsGrid<T>& grid, const std::vector<uint64_t>& records, Mesh& mesh)
{
    mesh.Clear();
    AxisNormals(mesh);
    const int64_t n = static_cast<int64_t>(grid.VoxelsPerSide());
    const int64_t nv = n + 1;
    std::unordered_map<int64_t, uint32_t> vertexOf;
    auto vertex = [&](int64_t x, int64_t y, int64_t z) -> uint32_t {
        const int64_t key = (z * nv + y) * nv + x;
        auto [it, fresh] = vertexOf.try_emplace(key, static_cast<uint32_t>(mesh.Coords.size()));
        if (fresh)
            mesh.Coords.emplace_back(grid.OriginX() + (static_cast<unsigned>(x) * grid.VoxelSize()), grid.OriginY() + (static_cast<unsigned>(y) * grid.VoxelSize()),
                                     grid.OriginZ() + (static_cast<unsigned>(z) * grid.VoxelSize()));
        return it->second;
    };
    // plane: 0 = XY (normal Z), 2 = XZ (normal Y), 1 = YZ (normal X) -- the reference's plane_index (:31)
    auto face = [&](int64_t x, int64_t y, int64_t z, int plane, int front) {
        uint32_t q[4];
        for (int v = 0; v < 2; ++v)
            for (int u = 0; u < 2; ++u)
                q[u + 2 * v] = plane == 0 ? vertex(x + u, y + v, z + front) : plane == 2 ? vertex(x + u, y + front, z + v) : vertex(x + front, y + v, z + u);
        const bool flip = (front != 0) == (plane != 0);             // (:67-85)
        if (flip) mesh.FacesCoords.insert(mesh.FacesCoords.end(), {q[0], q[2], q[1], q[1], q[2], q[3]});
        else      mesh.FacesCoords.insert(mesh.FacesCoords.end(), {q[0], q[1], q[2], q[1], q[3], q[2]});
        mesh.FacesNormals.insert(mesh.FacesNormals.end(), 6, static_cast<uint32_t>(front * 3 + plane));
    };
    for (const uint64_t rec : records) {
        const int64_t idx = static_cast<int64_t>(rec & kIndexMask);
        const unsigned mask = static_cast<unsigned>(rec >> 40);     // bit axis * 2 + side: that neighbour is unset / outside
        const int64_t x = idx % n, y = (idx / n) % n, z = idx / (n * n);
        if ((mask >> 4) & 1u) face(x, y, z, 0, 0);                  // XY back: unless (x, y, z - 1) emitted it
        face(x, y, z, 0, 1);
        if ((mask >> 2) & 1u) face(x, y, z, 2, 0);                  // XZ back: (x, y - 1, z)
        face(x, y, z, 2, 1);
        if ((mask >> 0) & 1u) face(x, y, z, 1, 0);                  // YZ back: (x - 1, y, z)
        face(x, y, z, 1, 1);
    }
    mesh.Colors.assign(mesh.VerticesSize(), Color(1.0f, 1.0f, 1.0f, 1.0f));
}

// The visible surface alone (an option of this build, not a reference export): only the faces between a set voxel and an unset /
// outside neighbour, outward winding.
template <VGType T>
void EmitSurface(const VoxelsGrid<T>& grid, const std::vector<uint64_t>& records, Mesh& mesh)
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
    // the two in-plane axes of each face direction, ordered so that (u x v) points along +axis
    static const int U[3][3] = {{0, 1, 0}, {0, 0, 1}, {1, 0, 0}};   // axis X: u = Y ; axis Y: u = Z ; axis Z: u = X
    static const int V[3][3] = {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}};   // axis X: v = Z ; axis Y: v = X ; axis Z: v = Y
    static const uint32_t normalIndex[3][2] = {{5, 2}, {4, 1}, {3, 0}};   // [axis][positive side]
    for (const uint64_t rec : records) {
        const int64_t idx = static_cast<int64_t>(rec & kIndexMask);
        const unsigned mask = static_cast<unsigned>(rec >> 40);
        const int64_t x = idx % n, y = (idx / n) % n, z = idx / (n * n);
        for (int axis = 0; axis < 3; ++axis)
            for (int side = 0; side < 2; ++side) {
                if (!((mask >> (axis * 2 + side)) & 1u)) continue;                   // interior face: not visible
                const int64_t ax = axis == 0, ay = axis == 1, az = axis == 2;
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
}

template <VGType T>
void EmitCubes(const VoxelsGrid<T>& grid, const Grid<float>& sdf, const std::vector<uint64_t>& records, Mesh& mesh)
{
    mesh.Clear();
    AxisNormals(mesh);
    const size_t n = grid.VoxelsPerSide();
    const float vs = grid.VoxelSize();
    const float max = Diagonal(n * vs);
    // corner c = dx + 2 dy + 4 dz (:92-94); the reference's twelve triangles in its order BACK, FRONT, TOP, BOTTOM, RIGHT, LEFT and the normal
    // slot it gives each face (:107-163 -- its BACK face carries slot 0 = (0,0,1), its FRONT face slot 3: kept as they are, the files must match)
    static const uint32_t tris[12][3] = {{0, 2, 1}, {1, 2, 3}, {4, 5, 6}, {5, 7, 6}, {6, 3, 2}, {3, 6, 7}, {0, 1, 4}, {1, 5, 4}, {1, 3, 5}, {3, 7, 5}, {0, 4, 2}, {2, 4, 6}};
    static const uint32_t faceSlot[6] = {0, 3, 1, 4, 2, 5};
    uint32_t cubes = 0;
    for (const uint64_t rec : records) {
        const size_t idx = static_cast<size_t>(rec & kIndexMask);
        const size_t x = idx % n, y = (idx / n) % n, z = idx / (n * n);
        if (std::isinf(sdf(x, y, z))) continue;
        const auto [r, g, b] = SDFToRGB(std::sqrt(sdf(x, y, z)), max);
        for (int c = 0; c < 8; ++c) {
            mesh.Coords.emplace_back(grid.OriginX() + (x * vs) + (vs * (c & 1)), grid.OriginY() + (y * vs) + (vs * ((c >> 1) & 1)),
                                     grid.OriginZ() + (z * vs) + (vs * ((c >> 2) & 1)));
            mesh.Colors.emplace_back(r, g, b, 1.0f);
        }
        const uint32_t base = cubes * 8;
        for (int t = 0; t < 12; ++t) {
            mesh.FacesCoords.insert(mesh.FacesCoords.end(), {base + tris[t][0], base + tris[t][1], base + tris[t][2]});
            mesh.FacesNormals.insert(mesh.FacesNormals.end(), 3, faceSlot[t / 2]);
        }
        ++cubes;
    }
    mesh.ShrinkToFit();
}

template <VGType T>
void EmitPoints(const VoxelsGrid<T>& grid, const Grid<float>& sdf, const std::vector<uint64_t>& records, Mesh& mesh)
{
    mesh.Clear();
    const size_t n = grid.VoxelsPerSide();
    const float vs = grid.VoxelSize();
    const float max = Diagonal(n * vs);
    for (const uint64_t rec : records) {
        const size_t idx = static_cast<size_t>(rec & kIndexMask);
        const size_t x = idx % n, y = (idx / n) % n, z = idx / (n * n);
        mesh.Coords.emplace_back(grid.OriginX() + (x * vs) + (vs / 2), grid.OriginY() + (y * vs) + (vs / 2),
                                 grid.OriginZ() + (z * vs) + (vs / 2));
        const auto [r, g, b] = SDFToRGB(std::sqrt(sdf(x, y, z)), max);
        mesh.Colors.emplace_back(r, g, b, 1.0f);
    }
    mesh.ShrinkToFit();
}

}  // namespace

template <VGType T> bool VoxelsGridToMeshCompressed(const VoxelsGrid<T>& grid, Mesh& mesh) { EmitReferenceFaces(grid, HostRecords(grid, VP_EXTRACT_FACES), mesh); return true; }
template <VGType T> bool VoxelsGridToSurfaceMesh(const VoxelsGrid<T>& grid, Mesh& mesh) { EmitSurface(grid, HostRecords(grid, VP_EXTRACT_EXPOSED), mesh); return true; }
template <VGType T> bool VoxelsGridToMesh(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh) { EmitCubes(grid, sdf, HostRecords(grid, VP_EXTRACT_SET), mesh); return true; }
template <VGType T> bool VoxelsGridToPointCloud(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh) { EmitPoints(grid, sdf, HostRecords(grid, VP_EXTRACT_SET), mesh); return true; }
template <VGType T> bool VoxelsGridToMeshCompressedDevice(const VoxelsGrid<T>& grid, Mesh& mesh) { EmitReferenceFaces(grid, DeviceRecords(grid, VP_EXTRACT_FACES), mesh); return true; }
template <VGType T> bool VoxelsGridToSurfaceMeshDevice(const VoxelsGrid<T>& grid, Mesh& mesh) { EmitSurface(grid, DeviceRecords(grid, VP_EXTRACT_EXPOSED), mesh); return true; }
template <VGType T> bool VoxelsGridToMeshDevice(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh) { EmitCubes(grid, sdf, DeviceRecords(grid, VP_EXTRACT_SET), mesh); return true; }
template <VGType T> bool VoxelsGridToPointCloudDevice(const VoxelsGrid<T>& grid, const Grid<float>& sdf, Mesh& mesh) { EmitPoints(grid, sdf, DeviceRecords(grid, VP_EXTRACT_SET), mesh); return true; }

#define VP_INSTANTIATE(T)                                                                         \
    template bool VoxelsGridToMeshCompressed<T>(const VoxelsGrid<T>&, Mesh&);                     \
    template bool VoxelsGridToSurfaceMesh<T>(const VoxelsGrid<T>&, Mesh&);                        \
    template bool VoxelsGridToMesh<T>(const VoxelsGrid<T>&, const Grid<float>&, Mesh&);           \
    template bool VoxelsGridToPointCloud<T>(const VoxelsGrid<T>&, const Grid<float>&, Mesh&);     \
    template bool VoxelsGridToMeshCompressedDevice<T>(const VoxelsGrid<T>&, Mesh&);               \
    template bool VoxelsGridToSurfaceMeshDevice<T>(const VoxelsGrid<T>&, Mesh&);                  \
    template bool VoxelsGridToMeshDevice<T>(const VoxelsGrid<T>&, const Grid<float>&, Mesh&);     \
    template bool VoxelsGridToPointCloudDevice<T>(const VoxelsGrid<T>&, const Grid<float>&, Mesh&);
VP_INSTANTIATE(uint32_t)
VP_INSTANTIATE(uint64_t)
#undef VP_INSTANTIATE
