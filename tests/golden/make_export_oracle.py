"""Regenerates tests/golden/export_oracle.json: what oracle/oracle_export.c (the restatement of the reference's three grid exporters)
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
    # VoxelsGridToMesh / VoxelsGridToPointCloud over the oracle's own sdf: [cube vertices, cube triangles, FNV of the index buffer, of the
    # coordinates, of the R G B bytes, points, FNV of their coordinates, of their R G B bytes]
    s = O.jfa(w, n, vs, origin)
    cc, crgb, cf, _ = O.grid_to_mesh_cubes(w, s, n, vs, origin)
    pc, prgb = O.grid_to_point_cloud(w, s, n, vs, origin)
    out["%s@%d:sdf" % (name, n)] = [int(cc.shape[0]), int(cf.shape[0]), O.fnv(cf), O.fnv(cc), O.fnv(crgb), int(pc.shape[0]), O.fnv(pc), O.fnv(prgb)]
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "export_oracle.json"), "w"), indent=1)
