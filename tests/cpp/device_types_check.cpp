// User-style code against the device-side types of the vplib mirror, written the way code against the reference would be:
// VoxelsGrid<T, device> (voxels_grid.h:32), DeviceVoxelsGrid (:244), DeviceGrid (grid.h:167), CudaPtr copy semantics
// (cuda_ptr.h:42-63), CalculateBoundingBox<device>(std::span<Position>, ...) (bounding_box.h:22-61).
// Prints "ok <what>" lines; any failed check prints "FAIL ..." and exits non-zero.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <span>
#include <vector>

#include <bounding_box.h>
#include <device_ptr.h>
#include <grid/grid.h>
#include <grid/voxels_grid.h>
#include <jfa/jfa.h>
#include <mesh/mesh_io.h>
#include <vox/vox.h>

#define CHECK(cond, what) do { if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); std::exit(1); } std::printf("ok %s\n", what); } while (0)

template <typename T>
static bool same_words(const HostVoxelsGrid<T>& a, const HostVoxelsGrid<T>& b)
{
    return a.View().StorageSize() == b.View().StorageSize() &&
           std::memcmp(a.View().Data(), b.View().Data(), a.View().StorageSize() * sizeof(T)) == 0 &&
           a.View().VoxelSize() == b.View().VoxelSize() && a.View().OriginX() == b.View().OriginX() &&
           a.View().OriginY() == b.View().OriginY() && a.View().OriginZ() == b.View().OriginZ();
}

template <typename T>
static void voxel_grids(const Mesh& mesh, size_t n, float vs, const float o[3])
{
    HostVoxelsGrid<T> host(n, vs);
    host.View().SetOrigin(o[0], o[1], o[2]);
    VOX::Compute<Types::TILED>(32, host, mesh);
    VoxelsGrid<T, false>& hv = host.View();                     // the reference's spelling of a host view
    size_t set = 0;
    for (size_t z = 0; z < n; ++z) for (size_t y = 0; y < n; ++y) for (size_t x = 0; x < n; ++x) set += hv.Voxel(x, y, z) ? 1 : 0;
    CHECK(set > 0, "voxelized grid is not empty");

    DeviceVoxelsGrid<T> dev(host);                              // upload
    const VoxelsGrid<T, true>& dv = dev.View();                 // device view: frame + device pointer
    CHECK(dv.VoxelsPerSide() == n && dv.VoxelSize() == vs && dv.OriginY() == o[1] && dv.StorageSize() == hv.StorageSize() && dv.Data() != nullptr,
          "DeviceVoxelsGrid(const HostVoxelsGrid&) keeps the frame");
    DeviceVoxelsGrid<T> copy(dev);                              // device-to-device deep copy
    CHECK(copy.View().Data() != dev.View().Data(), "DeviceVoxelsGrid copy owns its own storage");
    HostVoxelsGrid<T> back(copy);                               // download
    CHECK(same_words(back, host), "Host -> Device -> Device copy -> Host round trip");

    DeviceVoxelsGrid<T> zero(n, vs);                            // zero-filled (voxels_grid.cu:73,83)
    HostVoxelsGrid<T> z(zero);
    bool allZero = true;
    for (size_t i = 0; i < z.View().StorageSize(); ++i) allZero = allZero && z.View().Data()[i] == 0;
    CHECK(allZero, "DeviceVoxelsGrid(n) is zero-filled");
    zero = dev;                                                 // copy assignment (same size: no re-allocation needed)
    DeviceVoxelsGrid<T> small(32, 1.0f);
    small = dev;                                                // copy assignment across sizes re-allocates
    CHECK(same_words(HostVoxelsGrid<T>(zero), host) && same_words(HostVoxelsGrid<T>(small), host), "DeviceVoxelsGrid copy assignment");
    DeviceVoxelsGrid<T> moved(std::move(small));
    CHECK(same_words(HostVoxelsGrid<T>(moved), host) && small.View().Data() == nullptr, "DeviceVoxelsGrid move leaves the source empty");
    swap(moved, zero);
    CHECK(same_words(HostVoxelsGrid<T>(moved), host), "swap(DeviceVoxelsGrid&, DeviceVoxelsGrid&)");
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    Mesh mesh;
    if (!ImportMesh(argv[1], mesh)) return 3;
    const size_t n = std::strtoul(argv[2], nullptr, 10);

    // ---- bounding box with the reference's exact argument type, host and device flavour
    std::pair<float, float> bx, by, bz;
    const float side = CalculateBoundingBox<false>(std::span<Position>(mesh.Coords), bx, by, bz);
    CudaPtr<Position> dpts(mesh.Coords.data(), mesh.Coords.size());
    std::pair<float, float> cx, cy, cz;
    const float side2 = CalculateBoundingBox<true>(std::span<Position>(dpts.get(), dpts.Size()), cx, cy, cz);
    CHECK(side == side2 && bx == cx && by == cy && bz == cz && side > 0, "CalculateBoundingBox<false> == CalculateBoundingBox<true>");
    const float vs = side / static_cast<unsigned>(n);
    const float o[3] = {bx.first, by.first, bz.first};

    // ---- CudaPtr value semantics
    std::vector<float> v(1000);
    for (size_t i = 0; i < v.size(); ++i) v[i] = 0.5f * static_cast<float>(i);
    CudaPtr<float> a(v.data(), v.size());
    CudaPtr<float> b(a);                                        // alloc + device-to-device copy
    CudaPtr<float> c;
    c = a;                                                      // assignment allocates (sizes differ) and copies
    a.SetMemoryToZero();                                        // the copies must not alias the source
    std::vector<float> vb(v.size()), vc(v.size()), va(v.size());
    b.CopyToHost(vb.data(), vb.size()); c.CopyToHost(vc.data(), vc.size()); a.CopyToHost(va.data(), va.size());
    CHECK(vb == v && vc == v && va == std::vector<float>(v.size(), 0.0f) && b.get() != a.get() && c.Size() == v.size(), "CudaPtr copy = deep device-to-device copy");
    bool threw = false;
    try { a.CopyToHost(va.data(), va.size() + 1); } catch (const std::out_of_range&) { threw = true; }
    CHECK(threw, "CudaPtr::CopyToHost throws std::out_of_range past the end");
    a.CopyFromHost(v.data(), 10);                               // re-allocates to the new size (cuda_ptr.h:79-86)
    CHECK(a.Size() == 10, "CudaPtr::CopyFromHost adopts the new size");

    // ---- dense grids
    HostGrid<float> hs(n, -INFINITY);
    hs.View()(1, 2, 3) = 42.0f;
    DeviceGrid<float> ds(hs);                                   // upload
    DeviceGrid<float> ds2 = ds;                                 // device-to-device
    HostGrid<float> hs2(ds2);                                   // download
    CHECK(hs2.View()(1, 2, 3) == 42.0f && std::isinf(hs2.View()(0, 0, 0)) && hs2.View().Size() == n * n * n && ds2.View().Data() != ds.View().Data(),
          "HostGrid -> DeviceGrid -> DeviceGrid copy -> HostGrid");
    DeviceGrid<Position> dp(n);                                 // uninitialised device storage of a given size (grid.h:175)
    CHECK(dp.View().SizeX() == n && dp.View().Data() != nullptr, "DeviceGrid<Position>(size)");

    voxel_grids<uint32_t>(mesh, n, vs, o);
    voxel_grids<uint64_t>(mesh, n, vs, o);

    // ---- the usual pipeline still reads the same with the aliases
    HostVoxelsGrid32bit g(n, vs);
    g.View().SetOrigin(o[0], o[1], o[2]);
    VOX::Compute<Types::TILED>(32, g, mesh);
    DeviceVoxelsGrid32bit onDevice(g);
    HostVoxelsGrid32bit fromDevice(onDevice);
    HostGrid<float> sdf(n, -INFINITY), sdf2(n, -INFINITY);
    JFA::Compute<Types::TILED>(g, sdf);
    JFA::Compute<Types::TILED>(fromDevice, sdf2);
    CHECK(std::memcmp(sdf.View().Data(), sdf2.View().Data(), sdf.View().Size() * sizeof(float)) == 0, "JFA on a grid that went through the device types");
    std::printf("done\n");
    return 0;
}
