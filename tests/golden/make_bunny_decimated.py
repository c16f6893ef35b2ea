"""Generates tests/golden/bunny_decimated_n64.json + .grid.u32: BASELINE config 1 ("bunny.obj (3,510 tris) N=64 sequential CPU
path (-t 0), solid voxelize only").  The reference's 3,510-face file is not in its repository; the stand-in is the seeded
vertex-cluster decimation of assets/bunny.obj (cuda_mesh_voxelization_amd/mesh.py: decimate_cluster).  The expected grid is
the output of the CPU oracle (oracle/vp_oracle.c, pinned to the reference by survey_table.json) on that mesh.

    python tests/golden/make_bunny_decimated.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cuda_mesh_voxelization_amd import mesh as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = 64
xyz, tri = M.bunny_decimated()
origin, vs = O.frame([xyz], n)
words = O.voxelize(xyz, tri, n, vs, origin)
words.tofile(os.path.join(HERE, "bunny_decimated_n64.grid.u32"))
json.dump({"_provenance": "oracle/vp_oracle.c on mesh.bunny_decimated(); see make_bunny_decimated.py",
           "vertices": int(xyz.shape[0]), "faces": int(tri.shape[0]), "n": n,
           "mesh_fnv": [O.fnv(xyz), O.fnv(tri)], "origin": [float(v) for v in origin], "voxel_size": float(vs),
           "grid": [O.popcount(words), O.fnv(words)]},
          open(os.path.join(HERE, "bunny_decimated_n64.json"), "w"), indent=1)
print(open(os.path.join(HERE, "bunny_decimated_n64.json")).read())
