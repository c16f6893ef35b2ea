"""Dev tool: step time (voxelize + JFA, n = 512) with and without the per-kernel hipEvents of vp_prof, interleaved."""
import sys, os, math, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.pipeline import Engine
from cuda_mesh_voxelization_amd.capi import Frame, ALGO_TILED
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dev = torch.device("cuda", 0)
dx = torch.from_numpy(xyz.copy()).to(dev); dt = torch.from_numpy(tri.astype("int32")).to(dev)
g = torch.zeros(fr.words, dtype=torch.int32, device=dev); sdf = torch.empty(fr.voxels, dtype=torch.float32, device=dev)
def steps(k):
    for _ in range(k):
        eng.ctx.voxelize(fr, g.data_ptr(), dx.data_ptr(), dx.shape[0], dt.data_ptr(), dt.shape[0], ALGO_TILED)
        eng.ctx.jfa(fr, g.data_ptr(), -math.inf, sdf.data_ptr())
def timed(k, prof):
    eng.ctx.prof_reset(); eng.ctx.prof_enable(prof)
    steps(2); torch.cuda.synchronize(); t = time.perf_counter(); steps(k); torch.cuda.synchronize(); e = (time.perf_counter() - t) / k
    eng.ctx.prof_enable(False); return e * 1e3
res = {True: [], False: []}
for r in range(8):
    for p in ((True, False) if r % 2 else (False, True)):
        res[p].append(timed(20, p))
for p in (True, False):
    print("prof %-5s median %.4f min %.4f ms/step" % (p, statistics.median(res[p]), min(res[p])))
