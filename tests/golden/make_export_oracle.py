"""Regenerates tests/golden/export_oracle.json: what oracle/oracle_export.c (the restatement of the reference's VoxelsGridToMeshCompressed)
produces for d20 / torus at n = 32.  These are the repository's OWN oracle outputs, recorded so that a change of the restatement shows."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from cuda_mesh_voxelization_amd import mesh as M
out = json.load(open(os.path.join(ROOT, "tests", "golden", "export_oracle.json")))
for name, n in (("d20.obj", 32), ("torus.obj", 32)):
    xyz, tri = M.import_mesh(M.asset(name)); origin, vs = O.frame([xyz], n); w = O.voxelize(xyz, tri, n, vs, origin)
    c, f, nn = O.grid_to_mesh_compressed(w, n, vs, origin)
    out["%s@%d" % (name, n)] = [int(c.shape[0]), int(f.shape[0]), O.fnv(f), O.fnv(nn), O.fnv(c)]
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "export_oracle.json"), "w"), indent=1)
