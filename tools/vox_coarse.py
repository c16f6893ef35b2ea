"""Dev tool: the tiled voxelizer on coarse meshes (every triangle large): per-kernel ms and the naive voxelizer beside it.
  python tools/vox_coarse.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
eng = Engine(0)
for name in ("d20.obj", "torus.obj", "sphere.obj", "bimba.obj"):
    xyz, tri = M.import_mesh(M.asset(name))
    d = eng.mesh_to_device(xyz, tri)
    for n in (512, 1024, 2048):
        origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
        g, h = eng.new_grid(fr), eng.new_grid(fr)
        res = {}
        for algo, nm in ((ALGO_TILED, "tiled"), (ALGO_NAIVE, "naive")):
            for _ in range(3): eng.voxelize(fr, d[0], d[1], out=g if algo == ALGO_TILED else h, algo=algo)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): eng.voxelize(fr, d[0], d[1], out=g if algo == ALGO_TILED else h, algo=algo)
            torch.cuda.synchronize(); res[nm] = (time.perf_counter() - t) / 10 * 1e3
        eng.ctx.prof_reset(); eng.ctx.prof_select(None); eng.ctx.prof_enable(True)
        for _ in range(5): eng.voxelize(fr, d[0], d[1], out=g, algo=ALGO_TILED)
        torch.cuda.synchronize(); eng.ctx.prof_enable(False)
        tab = {k: round(v["ms"] / 5, 4) for k, v in eng.ctx.prof().items()}
        print("%-10s %6d faces n=%4d  tiled %.4f ms  naive %.4f ms  equal %s  %s" % (name, tri.shape[0], n, res["tiled"], res["naive"], bool(torch.equal(g, h)), tab), flush=True)
        del g, h; torch.cuda.empty_cache()
