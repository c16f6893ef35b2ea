"""Dev tool (GPU box): how sparse the JFA state is before every pass (bunny x24, n = 512): voxels, 64-voxel row segments (= what a\nwave sees) and rows that hold a seed.  Decides what wave-uniform skipping of \"none\" candidates can save."""
import sys, os, math
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import Frame, ALGO_TILED
from cuda_mesh_voxelization_amd.pipeline import Engine
eng = Engine(0); n = 512
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
dx, dt = eng.mesh_to_device(xyz, tri); g = eng.voxelize(fr, dx, dt)
cur = torch.empty(fr.voxels, dtype=torch.int32, device=eng.device)
eng.ctx.jfa_init(fr, g.data_ptr(), None, None, cur.data_ptr())
none = 0xFF9FF200 - (1 << 32)                               # kNone9 (vp_internal.h) as int32
k = n // 2
while k >= 1:
    v = cur.view(n, n, n)
    isn = (v == none)
    seg = (~isn).view(n, n, n // 64, 64).any(-1)
    row = (~isn).view(n, n, n).any(-1)
    print("before k=%3d: non-none voxels %.3f  64-segments with a seed %.3f  rows with a seed %.3f" % (k, 1 - isn.float().mean().item(), seg.float().mean().item(), row.float().mean().item()))
    nxt = torch.empty_like(cur)
    eng.ctx.jfa_pass(fr, k, cur.data_ptr(), None, None, nxt.data_ptr(), ALGO_TILED)
    cur = nxt; k //= 2
