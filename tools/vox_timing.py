"""Dev tool: voxelizer device time for several meshes / sizes / algorithms (per-kernel via vp_prof)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
eng = Engine(0)
cases = [("d20.obj", 1), ("torus.obj", 1), ("sphere.obj", 1), ("bimba.obj", 1), ("bunny.obj", 1), ("bunny.obj", 24), ("bunny.obj", 192)]
for name, ref in cases:
    xyz, tri = M.refine(*M.import_mesh(M.asset(name)), ref)
    dx, dt = eng.mesh_to_device(xyz, tri)
    for n in (256, 512, 1024, 2048):
        origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
        g = eng.new_grid(fr)
        row = []
        for algo in (ALGO_TILED, ALGO_NAIVE):
            for _ in range(2): eng.voxelize(fr, dx, dt, out=g, algo=algo)
            eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
            for _ in range(5): eng.voxelize(fr, dx, dt, out=g, algo=algo)
            eng.ctx.prof_enable(False)
            p = eng.ctx.prof()
            tot = sum(v["ms"] for v in p.values()) / 5
            row.append("%s %.3f ms (%s)" % ("tiled" if algo == 2 else "naive", tot, " ".join("%s=%.3f" % (k[4:], v["ms"] / 5) for k, v in p.items())))
        print("%-11s x%-3d %8d tris n=%4d | %s | %s" % (name, ref, tri.shape[0], n, row[0], row[1]))
        del g
