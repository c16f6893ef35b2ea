"""oracle/_ref -- parts of the REFERENCE ITSELF, compiled from /root/reference where it lies (oracle/Makefile, target _ref: mesh/mesh_io.cpp,
mesh/grid_to_mesh.cpp, csg/sequential.cpp, grid/voxels_grid.cu, bounding_box.h behind oracle/ref_driver.cpp; NVIDIA's CUDA toolkit headers come
with the image, inside the triton wheel).  What it pins, by running the reference's own code on the same inputs:

  * the OBJ importer (ImportMesh) and the frame of a run (CalculateBoundingBox as apps/cli/main.cpp:65-87 uses it)   -> mesh.py, the oracle, vpcli
  * CSG::Compute<SEQUENTIAL> for the three operators                                                              -> the oracle, the golden table's CSG rows, the GPU kernel
  * VoxelsGridToMeshCompressed / VoxelsGridToMesh / VoxelsGridToPointCloud + ExportMesh                         -> oracle_export.c and the files vpcli -e / -s -e writes
  * the bit layout of VoxelsGrid (the exporters and CSG read the product's words through the reference's own Voxel() / Word())

The voxelizer and the JFA of the reference do NOT build here (vox.h needs <cub/cub.cuh>; jfa/sequential.cpp allocates with cudaMalloc): those two
stay pinned to the survey table (tests/golden/PROVENANCE.md).  The binary is built in the build container (where /root/reference exists) and
travels to the GPU box; without it these tests skip."""
import json
import os
import subprocess

import numpy as np
import pytest

from cuda_mesh_voxelization_amd import build, mesh as M
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = ["d20.obj", "torus.obj", "sphere.obj", "bunny.obj", "bimba.obj"]

pytestmark = pytest.mark.skipif(O.build_ref() is None, reason="oracle/_ref is not built (needs /root/reference: make -C oracle _ref)")


@pytest.fixture(scope="module")
def cli():
    return build.build_cli()


def _body(path):
    """an OBJ file without its comment lines (the two exporters sign their files differently)"""
    return [ln for ln in open(path) if not ln.startswith("#")]


@pytest.mark.parametrize("name", ASSETS)
def test_import_is_the_references_import(tmp_path, name):
    rx, rt = O.ref_import(M.asset(name), str(tmp_path))
    xyz, tri = M.import_mesh(M.asset(name))
    assert np.array_equal(rx.view(np.uint32), xyz.view(np.uint32)) and np.array_equal(rt, tri)


def test_import_of_the_generated_benchmark_mesh(tmp_path):
    """the refined bunny of the benchmark, written as an OBJ with nine significant digits and read back by the reference's importer: the floats
    the product voxelizes are the floats the reference would read"""
    xyz, tri = M.bunny(3)
    path = str(tmp_path / "bunny_x3.obj")
    with open(path, "w") as f:
        f.write("# Vertices: %d\n# Faces: %d\n" % (xyz.shape[0], tri.shape[0]))
        for v in xyz:
            f.write("v %.9g %.9g %.9g\n" % tuple(float(c) for c in v))
        f.write("vn 0 0 1\n")
        for t in tri:
            f.write("f %d//1 %d//1 %d//1\n" % tuple(int(i) + 1 for i in t))
    rx, rt = O.ref_import(path, str(tmp_path))
    assert np.array_equal(rx.view(np.uint32), xyz.view(np.uint32)) and np.array_equal(rt, tri.astype(np.uint32))
    mx, mt = M.import_mesh(path)
    assert np.array_equal(mx.view(np.uint32), rx.view(np.uint32)) and np.array_equal(mt, rt)


@pytest.mark.parametrize("names,n", [(["d20.obj"], 32), (["bunny.obj"], 64), (["bimba.obj", "bunny.obj"], 64), (["sphere.obj", "torus.obj"], 48), (["bimba.obj", "bunny.obj"], 512)])
def test_frame_is_the_references_frame(names, n):
    """origin = minima of the bounding box of ALL meshes, voxel size = its longest side / n (apps/cli/main.cpp:65-87; bounding_box.h:22-61 with
    its else-if): the reference's code, the oracle's restatement and the harness's mesh.frame give the same four floats"""
    ro, rv = O.ref_frame([M.asset(a) for a in names], n)
    meshes = [M.import_mesh(M.asset(a))[0] for a in names]
    oo, ov = O.frame(meshes, n)
    mo, mv = M.frame(meshes, n)
    for o, v in ((oo, ov), (mo, mv)):
        assert np.array_equal(np.asarray(o, np.float32).view(np.uint32), ro.view(np.uint32))
        assert np.float32(v).view(np.uint32) == rv.view(np.uint32)


@pytest.mark.parametrize("op", [1, 2, 3])
def test_csg_is_the_references_csg(tmp_path, op):
    """CSG::Compute<SEQUENTIAL> of the reference on the two grids of the golden table's bimba / bunny rows at n = 64: equal to the oracle word
    for word, and its hash IS the golden row's (tests/golden/survey_table.json) -- the table's CSG column re-derived from the reference itself"""
    n = 64
    a_xyz, a_tri = M.import_mesh(M.asset("bimba.obj"))
    b_xyz, b_tri = M.import_mesh(M.asset("bunny.obj"))
    origin, vs = O.frame([a_xyz, b_xyz], n)
    a, b = O.voxelize(a_xyz, a_tri, n, vs, origin), O.voxelize(b_xyz, b_tri, n, vs, origin)
    got = O.ref_csg(a, b, n, op, str(tmp_path))
    want = a.copy()
    O.csg(want, b.copy(), op)
    assert np.array_equal(got, want)
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_table.json")))["rows"]
    row = [r for r in rows if r["meshes"] == ["bimba.obj", "bunny.obj"] and r["n"] == n and r["op"] == op][0]
    assert [int(np.unpackbits(got.view(np.uint8)).sum()), O.fnv(got)] == row["csg"][:2]
    rng = np.random.default_rng(op)                                  # and on noise (every word pattern)
    x, y = rng.integers(0, 2**32, 32 * 32 * 32 // 32, dtype=np.uint32), rng.integers(0, 2**32, 32 * 32 * 32 // 32, dtype=np.uint32)
    want = x.copy()
    O.csg(want, y.copy(), op)
    assert np.array_equal(O.ref_csg(x, y, 32, op, str(tmp_path)), want)


@pytest.mark.parametrize("name,n", [("d20.obj", 32), ("torus.obj", 32), ("sphere.obj", 48)])
def test_oracle_exporters_are_the_references_exporters(tmp_path, name, n):
    """oracle/oracle_export.c (the restatements the GPU-side tests use) against the files the reference's own exporters write: vertex lines
    `v x y z r g b` and face lines `f a//n b//n c//n`, all three exports"""
    xyz, tri = M.import_mesh(M.asset(name))
    origin, vs = O.frame([xyz], n)
    words = O.voxelize(xyz, tri, n, vs, origin)
    sdf = O.jfa(words, n, vs, origin)
    comp, cubes, points = O.ref_export(words, sdf, n, vs, origin, str(tmp_path))

    def parse(path):
        v, c, f, fn = [], [], [], []
        for ln in open(path):
            p = ln.split()
            if ln.startswith("v "):
                v.append(p[1:4]); c.append(p[4:7])
            elif ln.startswith("f "):
                q = [t.split("//") for t in p[1:4]]
                f.append([int(t[0]) - 1 for t in q]); fn.append([int(t[1]) - 1 for t in q])
        return v, c, np.array(f, np.uint32).reshape(-1, 3), np.array(fn, np.uint32).reshape(-1, 3)

    fmt = lambda a: [["%.6f" % float(x) for x in row] for row in a]
    col = lambda rgb: [["%.6f" % float(np.float32(x) / np.float32(255.0)) for x in row] for row in rgb]
    co, fa, no = O.grid_to_mesh_compressed(words, n, vs, origin)
    v, c, f, fn = parse(comp)
    assert v == fmt(co) and np.array_equal(f, fa) and np.array_equal(fn, no) and all(x == ["1.000000"] * 3 for x in c)
    co, rgb, fa, no = O.grid_to_mesh_cubes(words, sdf, n, vs, origin)
    v, c, f, fn = parse(cubes)
    assert v == fmt(co) and c == col(rgb) and np.array_equal(f, fa) and np.array_equal(fn, no)
    co, rgb = O.grid_to_point_cloud(words, sdf, n, vs, origin)
    v, c, f, _ = parse(points)
    assert v == fmt(co) and c == col(rgb) and f.shape[0] == 0


@pytest.mark.parametrize("t,typename", [(0, "sequential"), (3, "openmp")])
@pytest.mark.parametrize("name,n", [("d20.obj", 32), ("torus.obj", 32)])
def test_cli_exports_are_the_references_files(cli, tmp_path, name, n, t, typename):
    """`vpcli <mesh> -n n -t 0|3 -s -e` against the reference's exporters run on the product's OWN grid and sdf (the -d dump): every line of the
    three OBJ files but the comment header"""
    p = subprocess.run([cli, M.asset(name), "-n", str(n), "-t", str(t), "-s", "-e", "-d", str(tmp_path / "d")], capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    words = np.fromfile(str(tmp_path / "d.grid.u32"), np.uint32)
    sdf = np.fromfile(str(tmp_path / "d.sdf.f32"), np.float32)
    ro, rv = O.ref_frame([M.asset(name)], n)
    comp, cubes, points = O.ref_export(words, sdf, n, rv, ro, str(tmp_path))
    out = tmp_path / "out"
    assert _body(out / ("%s_%s" % (typename, name))) == _body(comp)
    assert _body(out / ("sdf_%s_out.obj" % typename)) == _body(cubes)
    assert _body(out / ("sdf_point_cloud_%s_out.obj" % typename)) == _body(points)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("torus.obj", 32), ("bunny.obj", 128)])
def test_cli_gpu_exports_are_the_references_files(cli, tmp_path, name, n):
    """-t 2: grid from the HIP voxelizer, sdf from the HIP JFA, the walk over the grid from vp_extract -- the three files against the reference's
    exporters run on that grid and sdf"""
    p = subprocess.run([cli, M.asset(name), "-n", str(n), "-t", "2", "-s", "-e", "-d", str(tmp_path / "d")], capture_output=True, text=True, cwd=tmp_path, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    words = np.fromfile(str(tmp_path / "d.grid.u32"), np.uint32)
    sdf = np.fromfile(str(tmp_path / "d.sdf.f32"), np.float32)
    ro, rv = O.ref_frame([M.asset(name)], n)
    comp, cubes, points = O.ref_export(words, sdf, n, rv, ro, str(tmp_path))
    out = tmp_path / "out"
    assert _body(out / ("tiled_%s" % name)) == _body(comp)
    assert _body(out / "sdf_tiled_out.obj") == _body(cubes)
    assert _body(out / "sdf_point_cloud_tiled_out.obj") == _body(points)


@pytest.mark.gpu
@pytest.mark.parametrize("op", [1, 2, 3])
def test_gpu_csg_is_the_references_csg(engine, tmp_path, op):
    """vp_csg (csg_words) against CSG::Compute<SEQUENTIAL> of the reference on noise and on the bimba / bunny grids"""
    n = 64
    rng = np.random.default_rng(10 + op)
    a = rng.integers(0, 2**32, n * n * n // 32, dtype=np.uint32)
    b = rng.integers(0, 2**32, n * n * n // 32, dtype=np.uint32)
    da, db = engine.to_device(a, np.uint32), engine.to_device(b, np.uint32)
    engine.csg(da, db, op)
    engine.sync()
    assert np.array_equal(engine.words_to_numpy(da), O.ref_csg(a, b, n, op, str(tmp_path)))


# ---------------------------------------------------------------------------------------------- header-level functions of the mirror
@pytest.fixture(scope="module")
def host_api_check(tmp_path_factory):
    """tests/cpp/host_api_check.cpp: the same commands answered by THIS repository's vplib mirror"""
    exe = str(tmp_path_factory.mktemp("cpp") / "host_api_check")
    pkg = os.path.join(ROOT, "cuda_mesh_voxelization_amd")
    build.build_lib()
    srcs = [os.path.join(pkg, "vplib", "src", f) for f in sorted(os.listdir(os.path.join(pkg, "vplib", "src"))) if f.endswith(".cpp")]
    subprocess.check_call(["g++", "-std=c++23", "-O2", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(pkg, "vplib", "include"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_api_check.cpp")] + srcs + ["-o", exe, "-L", pkg, "-lvphip", "-Wl,-rpath," + pkg])
    return exe


def _pairs(seed, m=20000):
    """pairs of positions as the JFA sees them: grid corners o + i vs of random frames (incl. far origins: cancellation), and free floats"""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    o = ((rng.random(3) - 0.5) * 200).astype(f32)
    vs = f32(10.0 ** rng.uniform(-3, 0))
    i, j = rng.integers(0, 2048, (m, 3)), rng.integers(0, 2048, (m, 3))
    grid = np.concatenate([o + i.astype(f32) * vs, o + j.astype(f32) * vs], 1).astype(f32)
    free = ((rng.random((m, 6)) - 0.5) * 10.0 ** rng.uniform(-3, 3, (m, 1))).astype(f32)
    return np.concatenate([grid, free], 0)


def test_distance_and_vector_ops_are_the_references(tmp_path, host_api_check):
    """JFA::CalculateDistance (jfa/jfa.h:19-20) and Vec3::Cross / Dot (mesh/mesh.h:114-126): the reference's header functions, this
    repository's mirror and the float32 expression the numpy test backend uses -- bit for bit on 40,000 pairs (grid corners of random
    frames, where the subtraction cancels, and free floats over six decades)"""
    p = _pairs(7)
    p.tofile(str(tmp_path / "pairs.f32"))
    for cmd, width in (("distance", 1), ("vec", 4)):
        O._ref(cmd, str(tmp_path / "pairs.f32"), str(tmp_path / ("ref_%s.f32" % cmd)))
        subprocess.check_call([host_api_check, cmd, str(tmp_path / "pairs.f32"), str(tmp_path / ("own_%s.f32" % cmd))])
        ref = np.fromfile(str(tmp_path / ("ref_%s.f32" % cmd)), np.float32)
        own = np.fromfile(str(tmp_path / ("own_%s.f32" % cmd)), np.float32)
        assert ref.size == p.shape[0] * width and np.array_equal(ref.view(np.uint32), own.view(np.uint32)), cmd
    a, b = p[:, :3], p[:, 3:]
    d = b - a
    want = ((d[:, 0] * d[:, 0]) + (d[:, 1] * d[:, 1])) + (d[:, 2] * d[:, 2])                   # float32 throughout, no contraction
    assert np.array_equal(np.fromfile(str(tmp_path / "ref_distance.f32"), np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32))


def test_helpers_timer_grammar_and_assert_line_are_the_references(host_api_check):
    """GetTypesString / NextPow2 / GetFilename (proc_utils.h:11-40); the `[Label]: <ms> ms` line of a Profiling scope (profiling.h:8-26: what
    scripts/benchmarks.py parses); the `[file:line] CPU Assert: message` line and exit code of cpuAssert (debug_utils.h:52-64)"""
    import re
    args = ["misc", "/a/b/c.obj", "plain.obj", "dir/"]
    assert O._ref(*args) == subprocess.run([host_api_check] + args, capture_output=True, text=True).stdout
    line = re.compile(r"^\[TiledJFA::Processing\]: \d+\.\d{6} ms\n$")
    assert line.match(O._ref("profile", "TiledJFA::Processing"))
    assert line.match(subprocess.run([host_api_check, "profile", "TiledJFA::Processing"], capture_output=True, text=True).stdout)
    exe = O.build_ref()
    r = subprocess.run([exe, "assert", "Number of GPUs must be 1..64"], capture_output=True, text=True)
    m = subprocess.run([host_api_check, "assert", "Number of GPUs must be 1..64"], capture_output=True, text=True)
    shape = re.compile(r"^\[[^\]:]+:\d+\] CPU Assert: Number of GPUs must be 1\.\.64$")
    assert r.returncode == m.returncode == 255 and shape.match(r.stdout) and shape.match(m.stdout)
