"""Dev tool: per-kernel time of every rank of a ghost-plane job on one GPU (vp_prof).  python tools/ghost_prof.py [n] [world]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
from cuda_mesh_voxelization_amd.slab import GhostSlabPipeline, HipSlabBackend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
for rank in range(world):
    pipe = GhostSlabPipeline(HipSlabBackend(eng), fr, rank, world)
    for _ in range(3): pipe.voxelize(dx, dt); pipe.jfa()
    eng.sync(); eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
    reps = 5
    for _ in range(reps): pipe.voxelize(dx, dt); pipe.jfa()
    eng.ctx.prof_enable(False); eng.sync()
    p = eng.ctx.prof()
    print("rank", rank, "regions", [(k, b1 - b0) for k, b0, b1 in pipe.regions])
    tot = 0
    for k, v in p.items():
        if v["launches"]:
            print("   %-12s launches/step %.1f  ms/step %.4f  avg %.4f" % (k, v["launches"] / reps, v["ms"] / reps, v["ms"] / v["launches"])); tot += v["ms"] / reps
    print("   sum %.4f" % tot)
    del pipe
