"""Dev tool: the cyclic phase of ONE rank of the transposed multi-GPU pipeline (phase A: border mask, fused start, every cyclic pass), for
rocprofv3 runs (counters of jfa_pass_dense<..., CYC> on a rank's share of the grid).   python tools/run_cyclic_rank.py <n> <world> <rank> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
from cuda_mesh_voxelization_amd.slab import HipSlabBackend, TransposeSlabPipeline
n, world, rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
xyz, tri = M.bunny(24 if n <= 1024 else 192); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
pipe = TransposeSlabPipeline(HipSlabBackend(eng), fr, rank, world, None)
pipe.voxelize(dx, dt, algo=ALGO_TILED)
for _ in range(reps):
    pipe.phase_a()
eng.sync()
print("done: n = %d, rank %d of %d, cyclic steps %s" % (n, rank, world, pipe.plan["cyclic"]))
