"""CPU-only checks of the host side: mesh input, benchmark-mesh generators, and that the C-ABI
library loads and exports every symbol include/vphip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from cuda_mesh_voxelization_amd import build, capi, mesh as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assets_counts():
    expect = {"bunny.obj": (28088, 56172), "bimba.obj": (23112, 46220), "d20.obj": (None, 20),
              "sphere.obj": (None, 1280), "torus.obj": (None, 576)}
    for name, (nv, nt) in expect.items():
        xyz, tri = M.import_mesh(M.asset(name))
        assert tri.shape == (nt, 3)
        if nv:
            assert xyz.shape == (nv, 3)
        assert tri.max() < xyz.shape[0]
        assert xyz.dtype == np.float32 and tri.dtype == np.uint32


def _edge_manifold(tri):
    t = tri.astype(np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], 0)
    key = np.minimum(e[:, 0], e[:, 1]) * (t.max() + 1) + np.maximum(e[:, 0], e[:, 1])
    _, cnt = np.unique(key, return_counts=True)
    return cnt


def test_refine_counts_and_watertight():
    xyz, tri = M.import_mesh(M.asset("sphere.obj"))
    base = _edge_manifold(tri)
    for factor in (3, 4, 6, 24):
        x2, t2 = M.refine(xyz, tri, factor)
        assert t2.shape[0] == tri.shape[0] * factor
        assert t2.max() == x2.shape[0] - 1
        cnt = _edge_manifold(t2)
        assert cnt.max() == base.max() and cnt.min() == base.min()     # shared edges stay shared
        # original vertices are untouched => same frame
        assert np.array_equal(x2[: xyz.shape[0]], xyz)


def test_benchmark_mesh_sizes():
    # face counts of the reference's benchmark folders (benchmarks/benchmarks_v2/bunny_<faces>)
    _, t3 = M.bunny(3)
    assert t3.shape[0] == 168516
    _, t24 = M.bunny(24)
    assert t24.shape[0] == 1348128


def test_export_import_roundtrip(tmp_path):
    xyz, tri = M.refine(*M.import_mesh(M.asset("torus.obj")), 3)
    p = str(tmp_path / "t.obj")
    M.export_obj(p, xyz, tri)
    x2, t2 = M.import_mesh(p)
    assert np.array_equal(x2, xyz) and np.array_equal(t2, tri)


def test_header_symbols_all_exported():
    build.build_lib()
    hdr = open(os.path.join(ROOT, "include", "vphip.h")).read()
    declared = set(re.findall(r"\b(vp_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"vp_frame", "vp_ctx"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = ctypes.CDLL(capi.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert capi.lib().vp_abi_version() == 6


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.VPError):
        capi.Context(0)
    from cuda_mesh_voxelization_amd.pipeline import Engine
    with pytest.raises(RuntimeError):
        Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cuda_mesh_voxelization_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                # the product path must never reach the CPU oracle: no import, include, dlopen or mention of it at all
                assert "oracle" not in txt.lower(), os.path.join(dp, f)


def test_bench_bare_multi_gpu_invocation_starts_ranks():
    """`python bench.py --gpus 2` without a launcher starts its own ranks (bench.launch_ranks) instead of exiting with "must be launched
    with torch.distributed.run" (VERDICT r04 #1).  No GPU here: the RANKS fail loudly ("needs a GPU"), the parent forwards their exit code
    and prints no JSON line -- which shows the parent itself never asked for a device."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_slab_gpu.py::test_bench_bare_invocation_launches_its_own_ranks runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid-n", "256"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "must be launched" not in r.stderr and "bench.py needs a GPU" in r.stderr, r.stderr[-3000:]
    assert r.stdout.strip() == ""


def test_bench_argument_surface():
    """bench.py's contract flags (--gpus / --steps / --warmup) parse before torch is imported; --n and --grid-n are one option"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.parse_args(["--gpus", "4", "--steps", "7", "--warmup", "2", "--grid-n", "1024"])
    assert (a.gpus, a.steps, a.warmup, a.n, a.multi) == (4, 7, 2, 1024, "ghost")
    assert mod.parse_args([]).gpus == 1 and mod.parse_args(["--n", "2048"]).n == 2048
