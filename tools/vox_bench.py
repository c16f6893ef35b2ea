"""GPU box: device time of the voxelizer per kernel, tiled vs naive, over meshes with small and with large triangles.
  python tools/vox_bench.py [n ...]        (default 512 1024)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
eng = Engine(0)
sizes = [int(s) for s in sys.argv[1:]] or [512, 1024]
meshes = [("d20.obj", 1), ("torus.obj", 1), ("sphere.obj", 1), ("bunny.obj", 1), ("bunny.obj", 24)]
print("%-16s %9s %5s | %-62s | %s" % ("mesh", "faces", "n", "tiled: setup + scan + scatter + tile + fill = total (ms)", "naive (ms)"))
for name, ref in meshes:
    xyz, tri = M.import_mesh(M.asset(name))
    if ref > 1:
        xyz, tri = M.refine(xyz, tri, ref)
    dx, dt = eng.mesh_to_device(xyz, tri)
    for n in sizes:
        origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
        g = eng.new_grid(fr)
        res = {}
        for algo in (ALGO_TILED, ALGO_NAIVE):
            for _ in range(3):
                eng.voxelize(fr, dx, dt, out=g, algo=algo)
            eng.sync(); eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
            for _ in range(10):
                eng.voxelize(fr, dx, dt, out=g, algo=algo)
            eng.ctx.prof_enable(False)
            res[algo] = {k: v["ms"] / 10 for k, v in eng.ctx.prof().items()}
            res[(algo, "words")] = eng.words_to_numpy(g).copy()
        assert np.array_equal(res[(ALGO_TILED, "words")], res[(ALGO_NAIVE, "words")])
        t = res[ALGO_TILED]
        parts = [t.get(k, 0.0) for k in ("vox_setup", "vox_scan", "vox_scatter", "vox_tile", "vox_fill")]
        print("%-16s %9d %5d | %8.4f + %6.4f + %6.4f + %8.4f + %6.4f = %8.4f %12s | %8.4f" %
              (name + ("x%d" % ref if ref > 1 else ""), tri.shape[0], n, *parts, sum(parts), "", sum(res[ALGO_NAIVE].values())), flush=True)
