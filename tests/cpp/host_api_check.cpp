// TEST INFRASTRUCTURE: the host-only members of the vplib mirror that the GPU checks never touch -- the debug dumps
// (VoxelsGrid::Print, Grid::Print / PrintValue: /root/reference/vplib/src/grid/voxels_grid.h:171-183, grid/grid.h:74-109) and the Color
// channel setters (mesh/mesh.h:19-36).  Prints to stdout; tests/test_cpp_api.py compares the text.
#include <cstdio>

#include "grid/grid.h"
#include "grid/voxels_grid.h"
#include "mesh/mesh.h"

int main()
{
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
