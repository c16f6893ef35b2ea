"""Dev tool: run voxelize + full JFA a few times (for rocprofv3 runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.new_grid(fr); sdf = torch.empty(fr.voxels, dtype=torch.float32, device=eng.device)
for _ in range(reps):
    eng.voxelize(fr, dx, dt, out=g, algo=ALGO_TILED)
    eng.jfa(fr, g, out=sdf, algo=ALGO_TILED)
eng.sync()
print("done")
