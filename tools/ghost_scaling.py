"""Dev tool: predicted strong scaling of the ghost-plane slab pipeline -- every rank of a G-GPU job is
run on THIS one GPU and timed; the job time on G GPUs is the slowest rank (no communication exists)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
from cuda_mesh_voxelization_amd.slab import GhostSlabPipeline, HipSlabBackend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.new_grid(fr); sdf = torch.empty(fr.voxels, dtype=torch.float32, device=eng.device)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
def single():
    eng.voxelize(fr, dx, dt, out=g); eng.jfa(fr, g, out=sdf)
t1 = timeit(single); print("1 gpu: %.3f ms" % t1)
for world in (2, 4, 8):
    ts = []
    for r in range(world):
        pipe = GhostSlabPipeline(HipSlabBackend(eng), fr, r, world)
        def step(): pipe.voxelize(dx, dt); pipe.jfa()
        ts.append(timeit(step, 5)); del pipe
    print("%d gpus: per-rank ms %s -> job %.3f ms, speedup %.2fx (planes/rank %s)" % (world, ["%.2f" % t for t in ts], max(ts), t1 / max(ts), ""))
