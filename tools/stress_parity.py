"""Dev tool (GPU box): sweep meshes x grid sizes, tiled voxelizer vs oracle and tiled JFA vs naive JFA (bit-exact)."""
import sys, os, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
from oracle import oracle as O
eng = Engine(0)
sizes = [int(s) for s in sys.argv[1:]] or [96, 160, 224, 256, 288, 320, 384, 416, 448, 480, 512, 544, 640]
bad = 0
for path in sorted(glob.glob(os.path.join(os.path.dirname(M.asset("bunny.obj")), "*.obj"))):
    xyz, tri = M.import_mesh(path)
    for n in sizes:
        origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
        dx, dt = eng.mesh_to_device(xyz, tri)
        g = eng.voxelize(fr, dx, dt, algo=ALGO_TILED)
        okv = np.array_equal(eng.words_to_numpy(g), O.voxelize(xyz, tri, n, vs, origin))
        s_t = eng.jfa(fr, g, algo=ALGO_TILED).clone()
        s_n = eng.jfa(fr, g, algo=ALGO_NAIVE)
        okj = bool(torch.equal(s_t.view(torch.int32), s_n.view(torch.int32)))
        bad += (not okv) + (not okj)
        print("%-12s n=%4d vox %s jfa %s" % (os.path.basename(path), n, "ok" if okv else "MISMATCH", "ok" if okj else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
