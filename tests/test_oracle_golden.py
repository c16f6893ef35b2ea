"""Pins the CPU oracle (oracle/vp_oracle.c) to the outputs of the reference's own sequential
path recorded in tests/golden/survey_table.json."""
import os

import numpy as np
import pytest

from cuda_mesh_voxelization_amd import mesh as M
from oracle import oracle as O


def _run_row(row, with_sdf=True):
    ms = [M.import_mesh(M.asset(f)) for f in row["meshes"]]
    n = row["n"]
    origin, vs = O.frame([m[0] for m in ms], n)
    grids = [O.voxelize(xyz, tri, n, vs, origin) for xyz, tri in ms]
    for g, (pc, h) in zip(grids, row["grids"]):
        assert O.popcount(g) == pc
        assert O.fnv(g) == h
    g0 = grids[0]
    for g in grids[1:]:
        O.csg(g0, g, row["op"])
    if "csg" in row:
        assert O.popcount(g0) == row["csg"][0]
        assert O.fnv(g0) == row["csg"][1]
    if with_sdf and row.get("sdf"):
        s = O.jfa(g0, n, vs, origin)
        st = O.sdf_stats(s)
        exp = row["sdf"]
        assert st["zeros"] == exp["zeros"]
        assert st["pinf"] == 0 and st["ninf"] == 0
        assert st["sum_pos"] == pytest.approx(exp["sum_pos"], rel=1e-8)
        assert st["sum_neg"] == pytest.approx(exp["sum_neg"], rel=1e-8)
        assert O.fnv(s) == exp["fnv"]


def _rows(golden_rows, pred):
    return [r for r in golden_rows if pred(r)]


def test_small_rows_bitmask_csg_sdf(golden_rows):
    rows = _rows(golden_rows, lambda r: r["n"] <= 128)
    assert len(rows) == 8
    for r in rows:
        _run_row(r)


def test_bunny_256_bitmask_and_sdf(golden_rows):
    (r,) = _rows(golden_rows, lambda r: r["n"] == 256)
    _run_row(r)


def test_bunny_1024_bitmask(golden_rows):
    (r,) = _rows(golden_rows, lambda r: r["n"] == 1024)
    _run_row(r, with_sdf=False)


def test_config3_512_bitmask_and_csg(golden_rows):
    (r,) = _rows(golden_rows, lambda r: r["n"] == 512)
    _run_row(r, with_sdf=False)


@pytest.mark.slow
def test_config3_512_sdf(golden_rows):
    (r,) = _rows(golden_rows, lambda r: r["n"] == 512)
    _run_row(r)


def test_frame_matches_python_mirror():
    for files in (["bunny.obj"], ["bimba.obj", "bunny.obj"], ["d20.obj"]):
        ms = [M.import_mesh(M.asset(f))[0] for f in files]
        for n in (32, 64, 512, 1024):
            o1, v1 = O.frame(ms, n)
            o2, v2 = M.frame(ms, n)
            assert np.array_equal(o1, o2) and v1 == v2


def test_voxelize_accumulates_by_xor():
    xyz, tri = M.import_mesh(M.asset("sphere.obj"))
    origin, vs = O.frame([xyz], 32)
    g = O.voxelize(xyz, tri, 32, vs, origin)
    g2 = O.voxelize(xyz, tri, 32, vs, origin, words=g.copy())
    assert not g2.any()                      # second pass toggles everything back (sequential.cpp:57)


def test_empty_grid_sdf_keeps_fill():
    n = 32
    words = np.zeros(O.nwords(n), np.uint32)
    s = O.jfa(words, n, 1.0, np.zeros(3, np.float32))
    assert np.all(np.isneginf(s))


def test_config1_decimated_bunny_fixture():
    """BASELINE config 1: the oracle reproduces the committed n = 64 grid of the seeded 3,511-face decimation, and the mesh
    generator is deterministic (tests/golden/make_bunny_decimated.py)."""
    import json
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gdir, "bunny_decimated_n64.json")))
    exp = np.fromfile(os.path.join(gdir, "bunny_decimated_n64.grid.u32"), np.uint32)
    xyz, tri = M.bunny_decimated()
    assert [O.fnv(xyz), O.fnv(tri)] == meta["mesh_fnv"]
    origin, vs = O.frame([xyz], 64)
    got = O.voxelize(xyz, tri, 64, vs, origin)
    assert np.array_equal(got, exp) and (O.popcount(got), O.fnv(got)) == tuple(meta["grid"])
