"""ctypes loader for the CPU oracle (oracle/libvporacle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvporacle.so")

_u32p = ctypes.POINTER(ctypes.c_uint32)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("vp_oracle.c", "oracle_export.c", "vp_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvporacle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.vpo_bounding_box.restype = ctypes.c_float
        L.vpo_bounding_box.argtypes = [_f32p, ctypes.c_size_t, _f32p]
        L.vpo_frame.restype = None
        L.vpo_frame.argtypes = [_f32p, ctypes.c_size_t, ctypes.c_uint, _f32p, _f32p]
        L.vpo_voxelize.restype = None
        L.vpo_voxelize.argtypes = [_u32p, ctypes.c_uint, ctypes.c_float, _f32p, _f32p, _u32p, ctypes.c_size_t]
        L.vpo_csg.restype = None
        L.vpo_csg.argtypes = [_u32p, _u32p, ctypes.c_size_t, ctypes.c_int]
        L.vpo_jfa.restype = ctypes.c_int
        L.vpo_jfa.argtypes = [_u32p, ctypes.c_uint, ctypes.c_float, _f32p, _f32p, ctypes.c_int]
        L.vpo_threads.restype = ctypes.c_int
        L.vpo_fnv1a64.restype = ctypes.c_uint64
        L.vpo_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.vpo_popcount.restype = ctypes.c_uint64
        L.vpo_popcount.argtypes = [_u32p, ctypes.c_size_t]
        L.vpo_sdf_stats.restype = None
        L.vpo_sdf_stats.argtypes = [_f32p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64),
                                    ctypes.POINTER(ctypes.c_double), _f32p]
        L.vpo_grid_to_mesh_compressed.restype = ctypes.c_int
        L.vpo_grid_to_mesh_compressed.argtypes = [_u32p, ctypes.c_uint, ctypes.c_float, _f32p, ctypes.POINTER(_f32p), ctypes.POINTER(ctypes.c_size_t),
                                                  ctypes.POINTER(_u32p), ctypes.POINTER(_u32p), ctypes.POINTER(ctypes.c_size_t)]
        _u8p = ctypes.POINTER(ctypes.c_ubyte)
        L.vpo_grid_to_mesh_cubes.restype = ctypes.c_int
        L.vpo_grid_to_mesh_cubes.argtypes = [_u32p, _f32p, ctypes.c_uint, ctypes.c_float, _f32p, ctypes.POINTER(_f32p), ctypes.POINTER(_u8p),
                                             ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(_u32p), ctypes.POINTER(_u32p), ctypes.POINTER(ctypes.c_size_t)]
        L.vpo_grid_to_point_cloud.restype = ctypes.c_int
        L.vpo_grid_to_point_cloud.argtypes = [_u32p, _f32p, ctypes.c_uint, ctypes.c_float, _f32p, ctypes.POINTER(_f32p), ctypes.POINTER(_u8p), ctypes.POINTER(ctypes.c_size_t)]
        L.vpo_free.restype = None
        L.vpo_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _pf(a):
    return a.ctypes.data_as(_f32p)


def _pu(a):
    return a.ctypes.data_as(_u32p)


def nwords(n: int) -> int:
    return (n * n * n + 31) // 32


def frame(meshes_xyz, n: int):
    allv = _f32(np.concatenate([np.asarray(m, np.float32).reshape(-1, 3) for m in meshes_xyz], 0))
    origin = np.zeros(3, np.float32)
    vs = ctypes.c_float()
    lib().vpo_frame(_pf(allv), allv.shape[0], n, _pf(origin), ctypes.byref(vs))
    return origin, np.float32(vs.value)


def voxelize(xyz, tri, n: int, voxel_size, origin, words=None):
    """Sequential voxelization; XOR-accumulates into `words` (fresh zero grid when None)."""
    xyz = _f32(xyz)
    tri = _u32(tri)
    origin = _f32(origin)
    if words is None:
        words = np.zeros(nwords(n), np.uint32)
    assert words.dtype == np.uint32 and words.flags.c_contiguous and words.size == nwords(n)
    lib().vpo_voxelize(_pu(words), n, float(voxel_size), _pf(origin), _pf(xyz), _pu(tri), tri.shape[0])
    return words


def csg(a, b, op: int):
    assert a.dtype == np.uint32 and b.dtype == np.uint32 and a.size == b.size
    b = _u32(b)
    lib().vpo_csg(_pu(a), _pu(b), a.size, op)
    return a


def jfa(words, n: int, voxel_size, origin, fill=-np.inf, max_passes: int = -1):
    """Sequential JFA; returns float32 [n^3] signed squared distances (x fastest)."""
    words = _u32(words)
    origin = _f32(origin)
    sdf = np.full(n * n * n, fill, np.float32)
    rc = lib().vpo_jfa(_pu(words), n, float(voxel_size), _pf(origin), _pf(sdf), max_passes)
    if rc != 0:
        raise MemoryError("vpo_jfa allocation failed")
    return sdf


def threads() -> int:
    return int(lib().vpo_threads())


def fnv(a) -> str:
    a = np.ascontiguousarray(a)
    return "%016x" % lib().vpo_fnv1a64(a.ctypes.data_as(ctypes.c_void_p), a.nbytes)


def popcount(words) -> int:
    words = _u32(words)
    return int(lib().vpo_popcount(_pu(words), words.size))


def sdf_stats(sdf):
    sdf = _f32(sdf)
    counts = (ctypes.c_uint64 * 3)()
    sums = (ctypes.c_double * 2)()
    mm = (ctypes.c_float * 2)()
    lib().vpo_sdf_stats(_pf(sdf), sdf.size, counts, sums, mm)
    return {"zeros": int(counts[0]), "pinf": int(counts[1]), "ninf": int(counts[2]),
            "sum_pos": float(sums[0]), "sum_neg": float(sums[1]), "min": float(mm[0]), "max": float(mm[1])}


def grid_to_mesh_compressed(words, n: int, voxel_size, origin):
    """The reference's VoxelsGridToMeshCompressed (oracle_export.c): (coords float32 [V, 3], faces uint32 [F, 3], face normal indices
    uint32 [F, 3]) -- every face of every set voxel once, in the reference's order."""
    words = _u32(words)
    origin = _f32(origin)
    pc, pf, pn = _f32p(), _u32p(), _u32p()
    nv, ni = ctypes.c_size_t(), ctypes.c_size_t()
    rc = lib().vpo_grid_to_mesh_compressed(_pu(words), n, float(voxel_size), _pf(origin), ctypes.byref(pc), ctypes.byref(nv),
                                           ctypes.byref(pf), ctypes.byref(pn), ctypes.byref(ni))
    if rc != 0:
        raise MemoryError("vpo_grid_to_mesh_compressed allocation failed")
    try:
        coords = np.ctypeslib.as_array(pc, shape=(nv.value * 3,)).copy().reshape(-1, 3) if nv.value else np.zeros((0, 3), np.float32)
        faces = np.ctypeslib.as_array(pf, shape=(ni.value,)).copy().reshape(-1, 3) if ni.value else np.zeros((0, 3), np.uint32)
        normals = np.ctypeslib.as_array(pn, shape=(ni.value,)).copy().reshape(-1, 3) if ni.value else np.zeros((0, 3), np.uint32)
    finally:
        lib().vpo_free(pc); lib().vpo_free(pf); lib().vpo_free(pn)
    return coords, faces, normals


def grid_to_mesh_cubes(words, sdf, n: int, voxel_size, origin):
    """The reference's VoxelsGridToMesh (oracle_export.c): (coords float32 [V, 3], rgb uint8 [V, 3] = R(), G(), B() of the vertex colours,
    faces uint32 [F, 3], face normal slots uint32 [F, 3]) -- 8 vertices and 12 triangles per set voxel with a finite sdf."""
    words, sdf, origin = _u32(words), _f32(sdf), _f32(origin)
    u8p = ctypes.POINTER(ctypes.c_ubyte)
    pc, pr, pf, pn = _f32p(), u8p(), _u32p(), _u32p()
    nv, ni = ctypes.c_size_t(), ctypes.c_size_t()
    rc = lib().vpo_grid_to_mesh_cubes(_pu(words), _pf(sdf), n, float(voxel_size), _pf(origin), ctypes.byref(pc), ctypes.byref(pr), ctypes.byref(nv),
                                      ctypes.byref(pf), ctypes.byref(pn), ctypes.byref(ni))
    if rc != 0:
        raise MemoryError("vpo_grid_to_mesh_cubes allocation failed")
    try:
        coords = np.ctypeslib.as_array(pc, shape=(max(nv.value, 1) * 3,)).copy()[:nv.value * 3].reshape(-1, 3)
        rgb = np.ctypeslib.as_array(pr, shape=(max(nv.value, 1) * 3,)).copy()[:nv.value * 3].reshape(-1, 3)
        faces = np.ctypeslib.as_array(pf, shape=(max(ni.value, 1),)).copy()[:ni.value].reshape(-1, 3)
        normals = np.ctypeslib.as_array(pn, shape=(max(ni.value, 1),)).copy()[:ni.value].reshape(-1, 3)
    finally:
        lib().vpo_free(pc); lib().vpo_free(pr); lib().vpo_free(pf); lib().vpo_free(pn)
    return coords, rgb, faces, normals


def grid_to_point_cloud(words, sdf, n: int, voxel_size, origin):
    """The reference's VoxelsGridToPointCloud (oracle_export.c): (coords float32 [V, 3], rgb uint8 [V, 3]), one vertex per set voxel."""
    words, sdf, origin = _u32(words), _f32(sdf), _f32(origin)
    u8p = ctypes.POINTER(ctypes.c_ubyte)
    pc, pr = _f32p(), u8p()
    nv = ctypes.c_size_t()
    rc = lib().vpo_grid_to_point_cloud(_pu(words), _pf(sdf), n, float(voxel_size), _pf(origin), ctypes.byref(pc), ctypes.byref(pr), ctypes.byref(nv))
    if rc != 0:
        raise MemoryError("vpo_grid_to_point_cloud allocation failed")
    try:
        coords = np.ctypeslib.as_array(pc, shape=(max(nv.value, 1) * 3,)).copy()[:nv.value * 3].reshape(-1, 3)
        rgb = np.ctypeslib.as_array(pr, shape=(max(nv.value, 1) * 3,)).copy()[:nv.value * 3].reshape(-1, 3)
    finally:
        lib().vpo_free(pc); lib().vpo_free(pr)
    return coords, rgb


# ---------------------------------------------------------------------------------------------- oracle/_ref: parts of the reference itself
# oracle/_ref/vpref (+ libvpref.so) = the reference's own mesh_io.cpp, grid_to_mesh.cpp, csg/sequential.cpp, voxels_grid.cu and
# bounding_box.h behind oracle/ref_driver.cpp, built by `make -C oracle _ref` from /root/reference where it lies (see oracle/Makefile for
# what can and what cannot be built in this image).  Built here (build container); on the GPU box only the prebuilt files are used.
_REF_DIR = os.path.join(_HERE, "_ref")
_REF_BIN = os.path.join(_REF_DIR, "vpref")
REFERENCE_ROOT = "/root/reference"


def build_ref(force: bool = False):
    """oracle/_ref/vpref, or None where it neither exists nor can be built (no /root/reference, no CUDA toolkit headers in the image)"""
    have = os.path.exists(_REF_BIN) and os.path.exists(os.path.join(_REF_DIR, "libvpref.so"))
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "vplib", "src")):
        srcs = [os.path.join(_HERE, "ref_driver.cpp"), os.path.join(_HERE, "Makefile")]
        stale = have and os.path.getmtime(_REF_BIN) < max(os.path.getmtime(f) for f in srcs)
        if force or stale or not have:
            r = subprocess.run(["make", "-C", _HERE] + (["-B"] if (force or stale) else []) + ["_ref"], capture_output=True, text=True)
            if r.returncode != 0:
                if have:
                    return _REF_BIN
                return None
            have = True
    return _REF_BIN if have else None


def _ref(*args, cwd=None):
    exe = build_ref()
    if exe is None:
        raise RuntimeError("oracle/_ref is not built (needs /root/reference and the CUDA toolkit headers of the image: make -C oracle _ref)")
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, cwd=cwd, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError("vpref %s failed (%d): %s%s" % (args[0], r.returncode, r.stdout[-500:], r.stderr[-500:]))
    return r.stdout


def ref_import(path, workdir):
    """the reference's ImportMesh (mesh/mesh_io.cpp:15-81): (Coords float32 [V, 3], FacesCoords uint32 [T, 3])"""
    prefix = os.path.join(workdir, "ref_import")
    _ref("import", path, prefix)
    return (np.fromfile(prefix + ".xyz.f32", np.float32).reshape(-1, 3), np.fromfile(prefix + ".tri.u32", np.uint32).reshape(-1, 3))


def ref_frame(paths, n):
    """the frame of a CLI run (apps/cli/main.cpp:65-87) through the reference's ImportMesh + CalculateBoundingBox: (origin float32 [3], voxel size float32)"""
    v = np.array([float(t) for t in _ref("frame", n, *paths).split()], np.float32)
    return v[:3].copy(), np.float32(v[3])


def ref_export(words, sdf, n, voxel_size, origin, workdir):
    """the reference's three exporters + ExportMesh: paths of <compressed>.obj, <cubes>.obj, <points>.obj (the last two None without an sdf)"""
    wp, sp, prefix = os.path.join(workdir, "ref_w.u32"), os.path.join(workdir, "ref_s.f32"), os.path.join(workdir, "ref")
    _u32(words).tofile(wp)
    if sdf is not None:
        _f32(sdf).tofile(sp)
    _ref("export", wp, sp if sdf is not None else "-", n, "%.9g" % float(voxel_size), *["%.9g" % float(c) for c in origin], prefix)
    return prefix + ".compressed.obj", (prefix + ".cubes.obj") if sdf is not None else None, (prefix + ".points.obj") if sdf is not None else None


def ref_csg(a, b, n, op, workdir):
    """the reference's CSG::Compute<Types::SEQUENTIAL> (csg/sequential.cpp:7-30): a op b as a new word array"""
    ap, bp, cp = (os.path.join(workdir, "ref_%s.u32" % k) for k in "abc")
    _u32(a).tofile(ap); _u32(b).tofile(bp)
    _ref("csg", ap, bp, n, op, cp)
    return np.fromfile(cp, np.uint32)
