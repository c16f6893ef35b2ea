"""Dev tool: border voxels per tile / per closed 4 x 4 x 4 lattice of jfa_first_two on the headline mesh (border mask from the GPU path).
  python tools/first_two_tiles.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.voxelize(fr, dx, dt)
bw = eng.surface(fr, g); eng.sync()
border = np.unpackbits(bw.cpu().numpy().view(np.uint8), bitorder="little").reshape(n, n, n).astype(bool)      # z, y, x
k = n // 4
XR = 16 if n <= 512 else 32
print("n = %d: %d border voxels (%.3f %%)" % (n, border.sum(), 100.0 * border.mean()))
cnt = border.reshape(4, k, 4, k, 4, k // XR, XR).sum(axis=(0, 2, 4, 6))       # per tile (rz, ry, x block)
print("tiles: %d, without a border voxel %.1f %%, mean %.2f, percentiles 10/50/90/99: %s" % (cnt.size, 100.0 * (cnt == 0).mean(), cnt.mean(), np.percentile(cnt, [10, 50, 90, 99])))
lat = border.reshape(4, k, 4, k, 4, k).sum(axis=(0, 2, 4))
print("closed lattices: none %.1f %%, one %.1f %%, more %.1f %%" % (100.0 * (lat == 0).mean(), 100.0 * (lat == 1).mean(), 100.0 * (lat > 1).mean()))
