// TEST INFRASTRUCTURE: the host-only members of the vplib mirror that the GPU checks never touch -- the debug dumps
// (VoxelsGrid::Print, Grid::Print / PrintValue: /root/reference/vplib/src/grid/voxels_grid.h:171-183, grid/grid.h:74-109) and the Color
// channel setters (mesh/mesh.h:19-36).  Prints to stdout; tests/test_cpp_api.py compares the text.
// With arguments it answers the commands of oracle/ref_driver.cpp (misc / distance / vec / profile / assert) with THIS library's functions,
// in the same output formats: tests/test_reference_build.py compares the two programs' outputs.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#ifndef PROFILING
#define PROFILING 1
#endif
#include "debug_utils.h"
#include "jfa/jfa.h"
#include "proc_utils.h"
#include "profiling.h"
#include "grid/grid.h"
#include "grid/voxels_grid.h"
#include "mesh/mesh.h"

static std::vector<float> ReadFloats(const char* path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<float> v(static_cast<size_t>(f.tellg()) / 4);
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * 4));
    return v;
}

static int Commands(int argc, char** argv)
{
    const std::string cmd = argv[1];
    if (cmd == "misc") {
        for (int t = 0; t < 4; ++t) std::printf("type %d %s\n", t, GetTypesString(static_cast<Types>(t)).c_str());
        const unsigned long ns[] = {0, 1, 2, 3, 31, 32, 33, 500, 512, 513, 1000, 100000};
        for (unsigned long n : ns) std::printf("nextpow2 %lu %lu %lu\n", n, NextPow2(n, 512), NextPow2(n, 1 << 20));
        for (int i = 2; i < argc; ++i) std::printf("filename %s\n", GetFilename(argv[i]).c_str());
        return 0;
    }
    if ((cmd == "distance" || cmd == "vec") && argc == 4) {
        const std::vector<float> p = ReadFloats(argv[2]);
        std::vector<float> r;
        for (size_t i = 0; i < p.size() / 6; ++i) {
            const Position a(p[i * 6], p[i * 6 + 1], p[i * 6 + 2]), b(p[i * 6 + 3], p[i * 6 + 4], p[i * 6 + 5]);
            if (cmd == "distance") r.push_back(JFA::CalculateDistance(a, b));
            else { const Position c = Position::Cross(a, b); r.insert(r.end(), {c.X, c.Y, c.Z, Position::Dot(a, b)}); }
        }
        std::ofstream(argv[3], std::ios::binary).write(reinterpret_cast<const char*>(r.data()), static_cast<std::streamsize>(r.size() * 4));
        return 0;
    }
    if (cmd == "profile" && argc == 3) { const std::string label = argv[2]; { PROFILING_SCOPE(label); } std::fflush(stdout); return 0; }
    if (cmd == "assert" && argc == 3) { cpuAssert(false, argv[2]); return 0; }
    return 2;
}

int main(int argc, char** argv)
{
    if (argc > 1) return Commands(argc, argv);
    HostVoxelsGrid<uint32_t> g(2, 1.0f);
    g.View().Voxel(1, 0, 0) = true;
    g.View().Voxel(0, 1, 1) = true;
    g.View().Print();
    std::printf("--\n");
    HostGrid<float> f(2, 0.5f);
    f.View()(1, 1, 0) = -2.25f;
    f.View().Print();
    std::printf("--\n");
    HostGrid<int> i(1, 2, 1, 7);
    i.View().Print();
    std::printf("--\n");
    HostGrid<Position> p(1, Position(1.0f, 2.0f, 3.5f));
    p.View().Print();
    std::printf("--\n");
    Color c(0.2f, 0.4f, 0.6f, 1.0f);
    std::printf("%u %u %u %u\n", c.R(), c.G(), c.B(), c.A());
    Color z(0.0f, 0.0f, 0.0f, 0.0f);
    z.R(1.0f);                                      // the other channels are 0: the setter's re-quantisation leaves them 0
    std::printf("%u %u %u %u\n", z.R(), z.G(), z.B(), z.A());
    return 0;
}
