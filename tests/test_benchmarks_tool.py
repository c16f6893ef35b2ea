"""tools/benchmarks.py reproduces the stdout -> CSV contract of the reference's scripts/benchmarks.py;
the expected column sets are the headers of the reference's committed CSVs
(benchmarks/benchmarks_v2/bunny_1348128/*.csv:1)."""
import csv
import os
import subprocess
import sys

import pytest

from cuda_mesh_voxelization_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "benchmarks.py")

# reference headers (data): benchmarks_v2/bunny_1348128/bunny_1348128_<variant>.csv line 1
REF_HEADERS = {
    "sequential_vox": ["size", "sequential_vox", "sequential_vox__processing"],
    "openmp_csg": ["size", "openmp_csg", "openmp_csg__processing"],
    "naive_vox": ["size", "naive_vox", "naive_vox__memory", "naive_vox__processing"],
    "naive_csg": ["size", "naive_csg", "naive_csg__memory", "naive_csg__processing"],
    "naive_jfa": ["size", "naive_jfa", "naive_jfa__initialization", "naive_jfa__memory", "naive_jfa__processing"],
    "tiled_vox": ["size", "tiled_vox", "tiled_vox__memory", "tiled_vox__processing",
                  "tiled_vox__tile_assignment__calculate_overlap", "tiled_vox__tile_assignment__compact_result",
                  "tiled_vox__tile_assignment__exclusive_scan", "tiled_vox__tile_assignment__work_queue_population",
                  "tiled_vox__tile_assignment__work_queue_sorting"],
}


def _run(tmp_path, types, sizes=(32, 64), niter=2):
    build.build_cli()
    out = tmp_path / "bench"
    subprocess.check_call([sys.executable, TOOL, "--niter", str(niter), "--minsize", str(sizes[0]), "--maxsize", str(sizes[-1]),
                           "--output", str(out), "--types"] + [str(t) for t in types] + ["--only", "d20.obj"],
                          stdout=subprocess.DEVNULL)
    res = {}
    for f in (out / "d20").iterdir():
        with open(f) as fh:
            rows = list(csv.reader(fh))
        res[f.stem[len("d20_"):]] = rows
    return res


def test_cpu_variants_csv_layout(tmp_path):
    res = _run(tmp_path, types=[3])          # -t 3: sequential vox + openmp csg / jfa (apps/cli/main.cpp:99-103,148-165,207-210)
    assert res["sequential_vox"][0] == REF_HEADERS["sequential_vox"]
    assert res["openmp_csg"][0] == REF_HEADERS["openmp_csg"]
    assert res["openmp_jfa"][0] == ["size", "openmp_jfa", "openmp_jfa__initialization", "openmp_jfa__memory", "openmp_jfa__processing"]
    for rows in res.values():
        body = rows[1:]
        assert [r[0] for r in body] == ["32", "32", "64", "64"]            # niter rows per size
        assert all(float(v) >= 0 for r in body for v in r[1:])


@pytest.mark.gpu
def test_gpu_variants_csv_layout(tmp_path):
    res = _run(tmp_path, types=[1, 2])
    for k in ("naive_vox", "naive_csg", "naive_jfa", "tiled_vox"):
        assert res[k][0] == REF_HEADERS[k], (k, res[k][0])
    assert res["tiled_jfa"][0] == ["size", "tiled_jfa", "tiled_jfa__initialization", "tiled_jfa__memory", "tiled_jfa__processing"]


@pytest.mark.skipif(not os.path.exists("/root/reference/scripts/benchmarks.py"), reason="the reference tree exists only in the build container")
def test_the_references_own_benchmark_script_drives_vpcli(tmp_path):
    """Drop-in at the outermost layer: the reference's scripts/benchmarks.py, UNCHANGED and run where it lies, with `vpcli` standing at the path
    it hard-wires for its executable (./build/Release/apps/cli/cli, scripts/benchmarks.py:38), the flags it passes (-n<N> -t<T> -m<iter> -p1 -s,
    :52-60) and its own parser of the `[Label]: <ms> ms` lines (:72-94).  CPU types here (-t 3: sequential voxelizer + OpenMP CSG / JFA,
    apps/cli/main.cpp:99-103); the CSV files it writes carry the headers of the reference's committed CSVs -- except the one column its own typo
    label `Openmp::Processing` (jfa/openmp.cpp:70) produces (`openmp__processing`; this build prints `OpenmpJFA::Processing`, the regular
    grammar its plot scripts subtract by name, scripts/plot_comparison.py:25-31)."""
    exe = tmp_path / "build" / "Release" / "apps" / "cli"
    exe.mkdir(parents=True)
    os.symlink(build.build_cli(), exe / "cli")
    assets = tmp_path / "tests"
    assets.mkdir()
    os.symlink(os.path.join(ROOT, "assets", "d20.obj"), assets / "d20.obj")
    r = subprocess.run([sys.executable, "/root/reference/scripts/benchmarks.py", "--niter", "2", "--minsize", "32", "--maxsize", "64", "--types", "3", "--output", "out"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = {}
    for f in (tmp_path / "out" / "d20").iterdir():
        with open(f) as fh:
            got[f.stem[len("d20_"):]] = list(csv.reader(fh))
    ref_dir = "/root/reference/benchmarks/benchmarks_v2/bunny_1348128"
    for variant in ("sequential_vox", "openmp_csg"):
        with open(os.path.join(ref_dir, "bunny_1348128_%s.csv" % variant)) as fh:
            assert got[variant][0] == next(csv.reader(fh)), variant               # the header the reference's own run produced
    with open(os.path.join(ref_dir, "bunny_1348128_openmp_jfa.csv")) as fh:
        ref_jfa = next(csv.reader(fh))
    assert sorted(c.replace("openmp__processing", "openmp_jfa__processing") for c in ref_jfa) == sorted(got["openmp_jfa"][0])
    for rows in got.values():
        assert [row[0] for row in rows[1:]] == ["32", "32", "64", "64"] and all(float(v) >= 0 for row in rows[1:] for v in row[1:])
