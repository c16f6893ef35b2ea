"""Dev tool: per JFA pass, the fraction of voxels whose state the pass leaves unchanged (the own voxel wins) -- what a gather
predicated on "winner != own" would skip.   python tools/own_winner_fraction.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.voxelize(fr, dx, dt)
cur = torch.empty(fr.voxels, dtype=torch.int32, device=eng.device); nxt = torch.empty_like(cur)
eng.ctx.jfa_init(fr, g.data_ptr(), None, None, cur.data_ptr())
k = n // 2
while k >= 1:
    eng.ctx.jfa_pass(fr, k, cur.data_ptr(), None, None, nxt.data_ptr(), ALGO_TILED); eng.sync()
    same = (cur == nxt)
    v = same.view(-1, 64)
    print("n = %d  k = %4d   unchanged voxels %5.1f %%   64-voxel row segments entirely unchanged %5.1f %%" % (n, k, 100.0 * same.float().mean().item(), 100.0 * v.all(dim=1).float().mean().item()))
    cur, nxt = nxt, cur; k //= 2
