// Exercises the C++ vplib mirror the way reference user code would (T = uint32_t and uint64_t,
// every Types value) and prints FNV-1a-64 hashes for the Python test to compare with the oracle.
//   api_check <obj> <n> <gpu:0|1> [slabs]      slabs > 1: the GPU variants once more on that many Z-slabs (vplib::SetDevices; the
//                                              contexts share device 0 here -- the code path is the multi-device one)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <span>
#include <vector>

#include <bounding_box.h>
#include <csg/csg.h>
#include <jfa/jfa.h>
#include <mesh/mesh_io.h>
#include <vox/vox.h>
#include <vp_runtime.h>

static uint64_t fnv(const void* p, size_t n)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

template <Types TY, typename T>
static void run(const char* tag, const Mesh& mesh, size_t n, float vs, const float o[3])
{
    HostVoxelsGrid<T> g(n, vs), h(n, vs);
    g.View().SetOrigin(o[0], o[1], o[2]);
    h.View().SetOrigin(o[0], o[1], o[2]);
    if constexpr (TY == Types::TILED) VOX::Compute<TY>(32, g, mesh); else VOX::Compute<TY>(g, mesh);
    std::printf("%s vox %016lx\n", tag, fnv(g.View().Data(), g.View().StorageSize() * sizeof(T)));
    // h = g with its lower half cleared; g \ h, g & h, g | h
    h = g;
    for (size_t z = 0; z < n / 2; ++z) for (size_t y = 0; y < n; ++y) for (size_t x = 0; x < n; ++x) h.View().Voxel(x, y, z) = false;
    constexpr Types CT = (TY == Types::TILED) ? Types::NAIVE : TY;
    HostVoxelsGrid<T> d = g; CSG::Compute<CT>(d, h, CSG::Difference<T>());
    HostVoxelsGrid<T> i = g; CSG::Compute<CT>(i, h, CSG::Intersection<T>());
    HostVoxelsGrid<T> u = d; CSG::Compute<CT>(u, i, CSG::Union<T>());
    std::printf("%s csg %016lx %016lx %016lx\n", tag, fnv(d.View().Data(), d.View().StorageSize() * sizeof(T)),
                fnv(i.View().Data(), i.View().StorageSize() * sizeof(T)), fnv(u.View().Data(), u.View().StorageSize() * sizeof(T)));
    HostGrid<float> sdf(n, -INFINITY);
    JFA::Compute<TY>(g, sdf);
    std::printf("%s sdf %016lx\n", tag, fnv(sdf.View().Data(), sdf.View().Size() * sizeof(float)));
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    Mesh mesh;
    if (!ImportMesh(argv[1], mesh)) return 3;
    const size_t n = std::strtoul(argv[2], nullptr, 10);
    const bool gpu = std::atoi(argv[3]) != 0;
    MinMax bx, by, bz;
    const float side = CalculateBoundingBox(std::span<const Position>(mesh.Coords.data(), mesh.Coords.size()), bx, by, bz);
    const float vs = side / static_cast<unsigned>(n);
    const float o[3] = {bx.first, by.first, bz.first};
    run<Types::SEQUENTIAL, uint32_t>("seq32", mesh, n, vs, o);
    run<Types::SEQUENTIAL, uint64_t>("seq64", mesh, n, vs, o);
    run<Types::OPENMP, uint32_t>("omp32", mesh, n, vs, o);
    if (gpu) {
        run<Types::NAIVE, uint32_t>("naive32", mesh, n, vs, o);
        run<Types::NAIVE, uint64_t>("naive64", mesh, n, vs, o);
        run<Types::TILED, uint32_t>("tiled32", mesh, n, vs, o);
        run<Types::TILED, uint64_t>("tiled64", mesh, n, vs, o);
        const int slabs = argc > 4 ? std::atoi(argv[4]) : 1;
        if (slabs > 1) {
            // what replaces the reference's cudaSetDevice(0) (apps/cli/main.cpp:22-23) for a library user with several GPUs
            vplib::SetDevices(std::vector<int>(static_cast<size_t>(slabs), 0), /*ghost=*/false);
            run<Types::TILED, uint32_t>("halo32", mesh, n, vs, o);
            run<Types::NAIVE, uint64_t>("halo64", mesh, n, vs, o);
            vplib::Shutdown();                                      // a new device list takes a new driver
            vplib::SetDevices(std::vector<int>(static_cast<size_t>(slabs), 0), /*ghost=*/true);
            run<Types::TILED, uint32_t>("ghost32", mesh, n, vs, o);
        }
    }
    return 0;
}
