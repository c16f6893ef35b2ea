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
