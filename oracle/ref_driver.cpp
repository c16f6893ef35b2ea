// ref_driver.cpp -- TEST INFRASTRUCTURE.  A command-line driver (this repository's own code) over functions of the REFERENCE compiled from
// its own sources where they lie under /root/reference (oracle/Makefile, target _ref; outputs only under oracle/_ref/):
//     vplib/src/mesh/mesh_io.cpp        ImportMesh, ExportMesh
//     vplib/src/mesh/grid_to_mesh.cpp   VoxelsGridToMeshCompressed, VoxelsGridToMesh, VoxelsGridToPointCloud (+ grid_to_mesh.h: SDFToRGB)
//     vplib/src/csg/sequential.cpp      CSG::Compute<Types::SEQUENTIAL, uint32_t, Union | Intersection | Difference>
//     vplib/src/grid/voxels_grid.cu     HostVoxelsGrid (host members only are ever called)
//     vplib/src/bounding_box.h          CalculateBoundingBox (header template)
// These translation units need <cuda_runtime.h> for the __host__ / __device__ decorations and types only: the image carries NVIDIA's own
// headers (the CUDA toolkit headers that ship inside the `triton` wheel); no CUDA library exists here, the CUDA runtime symbols that
// voxels_grid.cu references for its DEVICE classes stay unresolved and are never called.  What CANNOT be built this way: vox/*.cpp (vox.h
// includes <cub/cub.cuh>, absent) and jfa/sequential.cpp (allocates through cudaMalloc at run time, jfa/sequential.cpp:16) -- the voxelizer
// and the JFA stay pinned to the survey table only (tests/golden/PROVENANCE.md).
//
//   vpref import <in.obj> <out_prefix>                      -> <prefix>.xyz.f32, <prefix>.tri.u32 (FacesCoords), <prefix>.nrm.f32, <prefix>.fn.u32
//   vpref frame  <n> <a.obj> [<b.obj> ...]                  -> prints "ox oy oz voxel_size" as %.9g (apps/cli/main.cpp:65-87 through CalculateBoundingBox)
//   vpref export <words.u32> <sdf.f32|-> <n> <vs> <ox> <oy> <oz> <out_prefix>
//                                                           -> <prefix>.compressed.obj [, <prefix>.cubes.obj, <prefix>.points.obj] through ExportMesh
//   vpref csg    <a.u32> <b.u32> <n> <op 1|2|3> <out.u32>   -> CSG::Compute<SEQUENTIAL> (result in the first grid, csg/sequential.cpp:7-30)
//   vpref distance <pairs.f32> <out.f32>                    -> JFA::CalculateDistance (jfa/jfa.h:19-20) of every pair of positions (6 floats each)
//   vpref vec    <pairs.f32> <out.f32>                      -> Vec3::Cross and Vec3::Dot (mesh/mesh.h:114-126) of every pair: 4 floats each
//   vpref misc   [paths ...]                                -> GetTypesString(0..3), NextPow2 samples, GetFilename of every path (proc_utils.h:11-40)
//   vpref profile <label>                                   -> one Profiling scope (profiling.h:8-26): the timer line the benchmark script parses
//   vpref assert <message>                                  -> cpuAssert(false, message) (debug_utils.h:52-64)
//
// Two build parts (oracle/Makefile): the reference units + everything below as oracle/_ref/libvpref.so (undefined CUDA runtime symbols are
// legal in a shared object), and -DVPREF_LOADER: a main() that loads it with dlopen(RTLD_LAZY) -- lazy binding: a function that is never
// called is never resolved -- and forwards its arguments to vpref_main.
#ifdef VPREF_LOADER
#include <dlfcn.h>
#include <libgen.h>
#include <cstdio>
#include <string>
#include <unistd.h>
#include <limits.h>

int main(int argc, char** argv)
{
    char self[PATH_MAX] = {0};
    if (readlink("/proc/self/exe", self, sizeof(self) - 1) <= 0) { std::fprintf(stderr, "vpref: cannot find myself\n"); return 2; }
    const std::string lib = std::string(dirname(self)) + "/libvpref.so";
    void* h = dlopen(lib.c_str(), RTLD_LAZY | RTLD_LOCAL);
    if (!h) { std::fprintf(stderr, "vpref: %s\n", dlerror()); return 2; }
    auto fn = reinterpret_cast<int (*)(int, char**)>(dlsym(h, "vpref_main"));
    if (!fn) { std::fprintf(stderr, "vpref: %s\n", dlerror()); return 2; }
    return fn(argc, argv);
}
#else
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <bounding_box.h>
#include <csg/csg.h>
#include <jfa/jfa.h>
#include <profiling.h>
#include <proc_utils.h>
#include <debug_utils.h>
#include <grid/grid.h>
#include <grid/voxels_grid.h>
#include <mesh/grid_to_mesh.h>
#include <mesh/mesh.h>
#include <mesh/mesh_io.h>

namespace {

template <class T>
std::vector<T> ReadAll(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "vpref: cannot open %s\n", path.c_str()); std::exit(2); }
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    std::vector<T> v(static_cast<size_t>(bytes) / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(T)));
    return v;
}

template <class T>
void WriteAll(const std::string& path, const T* data, size_t count)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(count * sizeof(T)));
    if (!f) { std::fprintf(stderr, "vpref: cannot write %s\n", path.c_str()); std::exit(2); }
}

int Import(const std::string& in, const std::string& prefix)
{
    Mesh m;
    if (!ImportMesh(in, m)) { std::printf("import failed\n"); return 1; }
    static_assert(sizeof(Position) == 12, "Position is three floats");
    WriteAll(prefix + ".xyz.f32", reinterpret_cast<const float*>(m.Coords.data()), m.Coords.size() * 3);
    WriteAll(prefix + ".tri.u32", m.FacesCoords.data(), m.FacesCoords.size());
    WriteAll(prefix + ".nrm.f32", reinterpret_cast<const float*>(m.Normals.data()), m.Normals.size() * 3);
    WriteAll(prefix + ".fn.u32", m.FacesNormals.data(), m.FacesNormals.size());
    std::printf("vertices %zu indices %zu normals %zu\n", m.Coords.size(), m.FacesCoords.size(), m.Normals.size());
    return 0;
}

// apps/cli/main.cpp:65-87: the frame of a run = bounding box of the vertices of all meshes, voxel size = its longest side / n
int FrameOf(int n, int argc, char** argv)
{
    std::vector<Position> coords;
    for (int i = 0; i < argc; ++i) {
        Mesh m;
        if (!ImportMesh(argv[i], m)) { std::printf("import failed\n"); return 1; }
        coords.insert(coords.end(), m.Coords.begin(), m.Coords.end());
    }
    std::pair<float, float> bbX, bbY, bbZ;
    const float side = CalculateBoundingBox(std::span<Position>(&coords[0], coords.size()), bbX, bbY, bbZ);
    std::printf("%.9g %.9g %.9g %.9g\n", bbX.first, bbY.first, bbZ.first, side / n);
    return 0;
}

int Export(const std::string& wordsPath, const std::string& sdfPath, size_t n, float vs, float ox, float oy, float oz, const std::string& prefix)
{
    std::vector<uint32_t> words = ReadAll<uint32_t>(wordsPath);
    if (words.size() != VoxelsGrid<uint32_t>::CalculateStorageSize(n)) { std::fprintf(stderr, "vpref: %zu words for n = %zu\n", words.size(), n); return 2; }
    VoxelsGrid<uint32_t> grid(words.data(), n, vs);
    grid.SetOrigin(ox, oy, oz);
    Mesh out;
    if (!VoxelsGridToMeshCompressed(grid, out) || !ExportMesh(prefix + ".compressed.obj", out)) return 1;
    if (sdfPath != "-") {
        std::vector<float> sdf = ReadAll<float>(sdfPath);
        if (sdf.size() != n * n * n) { std::fprintf(stderr, "vpref: %zu sdf values for n = %zu\n", sdf.size(), n); return 2; }
        Grid<float> colors(sdf.data(), n);
        if (!VoxelsGridToMesh(grid, colors, out) || !ExportMesh(prefix + ".cubes.obj", out)) return 1;
        if (!VoxelsGridToPointCloud(grid, colors, out) || !ExportMesh(prefix + ".points.obj", out)) return 1;
    }
    return 0;
}

int Csg(const std::string& aPath, const std::string& bPath, size_t n, int op, const std::string& outPath)
{
    const std::vector<uint32_t> a = ReadAll<uint32_t>(aPath), b = ReadAll<uint32_t>(bPath);
    HostVoxelsGrid<uint32_t> ga(n, 1.0f), gb(n, 1.0f);
    const size_t words = VoxelsGrid<uint32_t>::CalculateStorageSize(n);
    if (a.size() != words || b.size() != words) { std::fprintf(stderr, "vpref: grids of %zu / %zu words for n = %zu\n", a.size(), b.size(), n); return 2; }
    std::memcpy(&ga.View().Word(0, 0, 0), a.data(), words * 4);
    std::memcpy(&gb.View().Word(0, 0, 0), b.data(), words * 4);
    switch (op) {                                                  // apps/cli/main.cpp:160-187
        case 1: CSG::Compute<Types::SEQUENTIAL, uint32_t>(ga, gb, CSG::Union<uint32_t>()); break;
        case 2: CSG::Compute<Types::SEQUENTIAL, uint32_t>(ga, gb, CSG::Intersection<uint32_t>()); break;
        case 3: CSG::Compute<Types::SEQUENTIAL, uint32_t>(ga, gb, CSG::Difference<uint32_t>()); break;
        default: std::fprintf(stderr, "vpref: op %d\n", op); return 2;
    }
    WriteAll(outPath, &ga.View().Word(0, 0, 0), words);
    return 0;
}

int Distance(const std::string& in, const std::string& out)
{
    const std::vector<float> p = ReadAll<float>(in);
    std::vector<float> d(p.size() / 6);
    for (size_t i = 0; i < d.size(); ++i)
        d[i] = JFA::CalculateDistance(Position(p[i * 6], p[i * 6 + 1], p[i * 6 + 2]), Position(p[i * 6 + 3], p[i * 6 + 4], p[i * 6 + 5]));
    WriteAll(out, d.data(), d.size());
    return 0;
}

int VecOps(const std::string& in, const std::string& out)
{
    const std::vector<float> p = ReadAll<float>(in);
    std::vector<float> r(p.size() / 6 * 4);
    for (size_t i = 0; i < p.size() / 6; ++i) {
        const Position a(p[i * 6], p[i * 6 + 1], p[i * 6 + 2]), b(p[i * 6 + 3], p[i * 6 + 4], p[i * 6 + 5]);
        const Position c = Position::Cross(a, b);
        r[i * 4] = c.X; r[i * 4 + 1] = c.Y; r[i * 4 + 2] = c.Z; r[i * 4 + 3] = Position::Dot(a, b);
    }
    WriteAll(out, r.data(), r.size());
    return 0;
}

int Misc(int argc, char** argv)
{
    for (int t = 0; t < 4; ++t) std::printf("type %d %s\n", t, GetTypesString(static_cast<Types>(t)).c_str());
    const unsigned long ns[] = {0, 1, 2, 3, 31, 32, 33, 500, 512, 513, 1000, 100000};
    for (unsigned long n : ns) std::printf("nextpow2 %lu %lu %lu\n", n, NextPow2(n, 512), NextPow2(n, 1 << 20));
    for (int i = 0; i < argc; ++i) std::printf("filename %s\n", GetFilename(argv[i]).c_str());
    return 0;
}

}  // namespace

extern "C" int vpref_main(int argc, char** argv)
{
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "import" && argc == 4) return Import(argv[2], argv[3]);
    if (cmd == "frame" && argc >= 4) return FrameOf(std::atoi(argv[2]), argc - 3, argv + 3);
    if (cmd == "export" && argc == 10)
        return Export(argv[2], argv[3], std::strtoull(argv[4], nullptr, 10), std::strtof(argv[5], nullptr), std::strtof(argv[6], nullptr),
                      std::strtof(argv[7], nullptr), std::strtof(argv[8], nullptr), argv[9]);
    if (cmd == "csg" && argc == 7) return Csg(argv[2], argv[3], std::strtoull(argv[4], nullptr, 10), std::atoi(argv[5]), argv[6]);
    if (cmd == "distance" && argc == 4) return Distance(argv[2], argv[3]);
    if (cmd == "vec" && argc == 4) return VecOps(argv[2], argv[3]);
    if (cmd == "misc") return Misc(argc - 2, argv + 2);
    if (cmd == "profile" && argc == 3) { { Profiling scope(argv[2]); } std::fflush(stdout); return 0; }
    if (cmd == "assert" && argc == 3) { cpuAssert(false, argv[2]); return 0; }
    std::fprintf(stderr, "usage: vpref import | frame | export | csg ... (see oracle/ref_driver.cpp)\n");
    return 2;
}
#endif  // VPREF_LOADER
