"""Host-side mesh helpers for the harness (tests, bench, smoke).

Mirrors, for the Python harness only, the reference's mesh input conventions:
  * ``import_mesh``  -- OBJ reader with the reference's accepted subset
    (``v x y z``, ``f a//b c//d e//f``; /root/reference/vplib/src/mesh/mesh_io.cpp:15-81).
    Coordinates are converted with libc ``strtof`` exactly like ``std::stof`` does.
  * ``frame``        -- shared grid frame of all input meshes
    (/root/reference/apps/cli/main.cpp:65-87, vplib/src/bounding_box.h:22-61).
  * deterministic subdivision generators that rebuild the reference's benchmark mesh
    sizes from assets/bunny.obj (56,172 faces): x3 -> 168,516, x24 -> 1,348,128,
    x192 -> 10,785,024.

The C++ mirror of the same API (used by the CLI) lives in vplib/.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import functools
import os

import numpy as np

_libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6")
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(REPO_ROOT, "assets")


def asset(name: str) -> str:
    return os.path.join(ASSETS, name)


@functools.lru_cache(maxsize=16)
def _import_mesh_cached(path: str, mtime: float):
    verts = []
    faces = []
    strtof = _libc.strtof
    with open(path, "rb") as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            tag = parts[0]
            if tag == b"v":
                verts.append((strtof(parts[1], None), strtof(parts[2], None), strtof(parts[3], None)))
            elif tag == b"f":
                # mesh_io.cpp:63-73: first three corners, leading integer of "a//b", 1-based
                faces.append(tuple(int(p.split(b"/")[0]) - 1 for p in parts[1:4]))
    xyz = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
    tri = np.asarray(faces, dtype=np.uint32).reshape(-1, 3)
    xyz.setflags(write=False)
    tri.setflags(write=False)
    return xyz, tri


def import_mesh(path: str):
    """Return (xyz float32 [V,3], tri uint32 [T,3]) for an OBJ file."""
    return _import_mesh_cached(path, os.path.getmtime(path))


def frame(meshes_xyz, n: int):
    """Grid frame over the concatenated vertices of all meshes (main.cpp:65-87).

    Returns (origin float32[3], voxel_size float32).  All arithmetic in float32.
    """
    allv = np.concatenate([np.asarray(m, dtype=np.float32).reshape(-1, 3) for m in meshes_xyz], axis=0)
    mn = allv.min(axis=0)
    mx = allv.max(axis=0)
    side = np.max((mx - mn).astype(np.float32))
    voxel = np.float32(side) / np.float32(n)
    return mn.astype(np.float32), np.float32(voxel)


# --------------------------------------------------------------------------- subdivision
def _edge_midpoints(xyz: np.ndarray, tri: np.ndarray):
    """Unique-edge midpoints (shared between the two faces of an edge => watertight)."""
    t = tri.astype(np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], axis=0)
    lo = np.minimum(e[:, 0], e[:, 1])
    hi = np.maximum(e[:, 0], e[:, 1])
    key = lo * np.int64(xyz.shape[0]) + hi
    uniq, inv = np.unique(key, return_inverse=True)
    ulo = (uniq // xyz.shape[0]).astype(np.int64)
    uhi = (uniq % xyz.shape[0]).astype(np.int64)
    mid = ((xyz[ulo] + xyz[uhi]) * np.float32(0.5)).astype(np.float32)
    T = tri.shape[0]
    m01 = inv[0:T] + xyz.shape[0]
    m12 = inv[T:2 * T] + xyz.shape[0]
    m20 = inv[2 * T:3 * T] + xyz.shape[0]
    return mid, m01, m12, m20


def subdivide_centroid(xyz, tri):
    """1 -> 3: insert the centroid of every face."""
    xyz = np.asarray(xyz, np.float32)
    t = np.asarray(tri).astype(np.int64)
    c = ((xyz[t[:, 0]] + xyz[t[:, 1]]) + xyz[t[:, 2]]) / np.float32(3.0)
    ci = np.arange(t.shape[0], dtype=np.int64) + xyz.shape[0]
    a, b, d = t[:, 0], t[:, 1], t[:, 2]
    out = np.stack([np.stack([a, b, ci], 1), np.stack([b, d, ci], 1), np.stack([d, a, ci], 1)], 1)
    return np.concatenate([xyz, c.astype(np.float32)], 0), out.reshape(-1, 3).astype(np.uint32)


def subdivide_midpoint(xyz, tri):
    """1 -> 4: split every edge at its midpoint."""
    xyz = np.asarray(xyz, np.float32)
    t = np.asarray(tri).astype(np.int64)
    mid, m01, m12, m20 = _edge_midpoints(xyz, t)
    a, b, c = t[:, 0], t[:, 1], t[:, 2]
    out = np.stack([
        np.stack([a, m01, m20], 1), np.stack([m01, b, m12], 1),
        np.stack([m20, m12, c], 1), np.stack([m01, m12, m20], 1)], 1)
    return np.concatenate([xyz, mid], 0), out.reshape(-1, 3).astype(np.uint32)


def subdivide_barycentric(xyz, tri):
    """1 -> 6: edge midpoints + centroid."""
    xyz = np.asarray(xyz, np.float32)
    t = np.asarray(tri).astype(np.int64)
    mid, m01, m12, m20 = _edge_midpoints(xyz, t)
    nv = xyz.shape[0] + mid.shape[0]
    cen = (((xyz[t[:, 0]] + xyz[t[:, 1]]) + xyz[t[:, 2]]) / np.float32(3.0)).astype(np.float32)
    ci = np.arange(t.shape[0], dtype=np.int64) + nv
    a, b, c = t[:, 0], t[:, 1], t[:, 2]
    out = np.stack([
        np.stack([a, m01, ci], 1), np.stack([m01, b, ci], 1),
        np.stack([b, m12, ci], 1), np.stack([m12, c, ci], 1),
        np.stack([c, m20, ci], 1), np.stack([m20, a, ci], 1)], 1)
    return np.concatenate([xyz, mid, cen], 0), out.reshape(-1, 3).astype(np.uint32)


_RECIPES = {
    1: (),
    3: ("centroid",),
    4: ("midpoint",),
    6: ("barycentric",),
    24: ("midpoint", "barycentric"),
    192: ("midpoint", "midpoint", "midpoint", "centroid"),
}
_STEPS = {"centroid": subdivide_centroid, "midpoint": subdivide_midpoint, "barycentric": subdivide_barycentric}


def refine(xyz, tri, factor: int):
    """Multiply the face count by `factor` (1, 3, 4, 6, 24 or 192) with the recipes of SURVEY 8(d)."""
    for step in _RECIPES[factor]:
        xyz, tri = _STEPS[step](xyz, tri)
    return np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(tri, np.uint32)


@functools.lru_cache(maxsize=4)
def bunny(factor: int = 1):
    """assets/bunny.obj refined to 56,172 * factor faces (24 -> the 1,348,128-face headline mesh)."""
    xyz, tri = import_mesh(asset("bunny.obj"))
    xyz, tri = refine(xyz, tri, factor)
    xyz.setflags(write=False)
    tri.setflags(write=False)
    return xyz, tri


def decimate_cluster(xyz, tri, cells: int = 24, seed: int = 80):
    """Seeded vertex-cluster decimation (the reference's 3,510-face bunny is not in its repo; this is the
    stand-in for BASELINE config 1).  Vertices are binned into a cells^3 lattice over the bounding cube, shifted by a
    seeded fraction of a cell; every cluster collapses to its float64 mean rounded to float32; faces that lose a corner
    and duplicate faces are dropped.  assets/bunny.obj, cells = 24, seed = 80 -> 1,747 vertices, 3,511 faces (the seed whose face count is closest to 3,510)."""
    xyz = np.asarray(xyz, np.float32)
    t = np.asarray(tri).astype(np.int64)
    off = np.random.default_rng(seed).random(3).astype(np.float32)
    mn = xyz.min(axis=0)
    side = np.float32((xyz.max(axis=0) - mn).max())
    q = np.floor((xyz - mn) / (side / np.float32(cells)) + off).astype(np.int64)
    key = (q[:, 2] * (cells + 2) + q[:, 1]) * (cells + 2) + q[:, 0]
    uniq, inv = np.unique(key, return_inverse=True)
    acc = np.zeros((uniq.size, 3), np.float64)
    np.add.at(acc, inv, xyz.astype(np.float64))
    cnt = np.bincount(inv, minlength=uniq.size)[:, None]
    nxyz = (acc / cnt).astype(np.float32)
    t = inv[t]
    t = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]
    srt = np.sort(t, axis=1)
    _, first = np.unique(srt[:, 0] * (1 << 40) + srt[:, 1] * (1 << 20) + srt[:, 2], return_index=True)
    t = t[np.sort(first)]
    return np.ascontiguousarray(nxyz, np.float32), np.ascontiguousarray(t, np.uint32)


@functools.lru_cache(maxsize=1)
def bunny_decimated():
    """BASELINE config 1 stand-in: assets/bunny.obj decimated to ~3.5 k faces (see decimate_cluster)."""
    xyz, tri = decimate_cluster(*import_mesh(asset("bunny.obj")))
    xyz.setflags(write=False)
    tri.setflags(write=False)
    return xyz, tri


def export_obj(path: str, xyz, tri):
    """Write `v`/`f a//a` lines with 9 significant digits (round-trips float32 through strtof)."""
    with open(path, "w") as f:
        f.write("# Vertices: %d\n# Faces: %d\n" % (len(xyz), len(tri)))
        for x, y, z in np.asarray(xyz, np.float32):
            f.write("v %.9g %.9g %.9g\n" % (x, y, z))
        for a, b, c in np.asarray(tri, np.int64) + 1:
            f.write("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c))
