"""Dev tool: time the full tiled JFA (vp_jfa) at a given n on one GPU: python tools/jfa_time.py 2048 [refine]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
refine = int(sys.argv[2]) if len(sys.argv) > 2 else 24
xyz, tri = M.bunny(refine); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.voxelize(fr, dx, dt)
sdf = torch.empty(fr.voxels, dtype=torch.float32, device=eng.device)
eng.jfa(fr, g, out=sdf, algo=ALGO_TILED); eng.sync()
for _ in range(2):
    t = time.perf_counter(); eng.jfa(fr, g, out=sdf, algo=ALGO_TILED); eng.sync()
    print("n %d jfa tiled %.1f ms" % (n, (time.perf_counter() - t) * 1e3))
print("checksum", int(sdf.view(torch.int32)[::4097].to(torch.int64).sum().item()))
