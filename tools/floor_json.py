"""profiles/formulation_floor.json from the floor microbenchmark's outputs of a round (tools/ubench/floor.hip -> <dir>/floor_n512.txt,
floor_n1024.txt; the last line of each is its JSON summary).   python tools/floor_json.py gpurun_out/r06 6 > profiles/formulation_floor.json"""
import json, os, sys

d, rnd = sys.argv[1], int(sys.argv[2])
out = {"source": "tools/ubench/floor.hip on the round-%d GPU box (profiles/r%02d/floor_n512.txt, floor_n1024.txt): one id volume in, one out, per voxel the "
                 "irreducible work of the exact 27-candidate pass (3 decodes, 3 x (sub, mul, 3 + 9 adds), 27 v_min_f64 on live data), nothing else" % (rnd, rnd),
       "round": rnd}
for n in (512, 1024):
    path = os.path.join(d, "floor_n%d.txt" % n)
    lines = [ln for ln in open(path) if ln.startswith("{")]
    out["n%d" % n] = json.loads(lines[-1])
print(json.dumps(out, indent=1))
