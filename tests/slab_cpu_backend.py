"""TEST INFRASTRUCTURE: a CPU stand-in for HipSlabBackend so that the Z-slab halo-exchange logic of
cuda_mesh_voxelization_amd/slab.py can run under gloo with world_size > 1 on a machine without GPUs.
Compute here is numpy float32 (no FMA) written from the reference semantics
(/root/reference/vplib/src/jfa/sequential.cpp:24-125); voxelization comes from the oracle."""
import numpy as np
import torch

from oracle import oracle as O

NONE = np.int32(-1)


def _np(t):
    return t.numpy()


class CpuSlabBackend:
    def __init__(self, mesh_host):
        self.mesh_host = mesh_host            # (xyz, tri) numpy: 'device' mesh handles are ignored

    def empty_u32(self, n):
        return torch.zeros(int(n), dtype=torch.int32)

    def empty_f32(self, n):
        return torch.zeros(int(n), dtype=torch.float32)

    def ids_u32(self, n):
        """id volumes of the ghost / hybrid pipelines (slab.py: filled once at allocation).  VP_SLAB_POISON = a voxel index every
        entry starts as: a (wrong) seed that would spoil the result if a plane nobody produced were ever consumed."""
        import os
        return torch.full((int(n),), int(os.environ.get("VP_SLAB_POISON", "0"), 0), dtype=torch.int32)

    # -- stages -------------------------------------------------------------------------------
    def voxelize(self, frame, words, d_xyz, d_tri, algo):
        xyz, tri = self.mesh_host
        n = frame.n
        full = O.voxelize(xyz, tri, n, frame.voxel_size, np.array(list(frame.origin), np.float32))
        pw = n * n // 32
        _np(words)[:] = full[frame.z0 * pw:frame.z1 * pw].view(np.int32)

    def csg(self, a, b, op):
        x = _np(a).view(np.uint32)
        y = _np(b).view(np.uint32)
        O.csg(x, y.copy(), op)

    @staticmethod
    def _bits(words_i32, n, planes):
        return np.unpackbits(words_i32.view(np.uint8), bitorder="little").reshape(planes, n, n).astype(bool)

    def jfa_init(self, frame, words, below, above, ids):
        n, nz = frame.n, frame.z1 - frame.z0
        occ = self._bits(_np(words), n, nz)
        lo = self._bits(_np(below), n, 1) if below is not None else np.zeros((1, n, n), bool)
        hi = self._bits(_np(above), n, 1) if above is not None else np.zeros((1, n, n), bool)
        vol = np.pad(np.concatenate([lo, occ, hi], 0), ((0, 0), (1, 1), (1, 1)))          # outside = unset
        interior = np.ones((nz, n, n), bool)
        for dz in range(3):
            for dy in range(3):
                for dx in range(3):
                    interior &= vol[dz:dz + nz, dy:dy + n, dx:dx + n]
        border = occ & ~interior
        zz, yy, xx = np.meshgrid(np.arange(frame.z0, frame.z1), np.arange(n), np.arange(n), indexing="ij")
        lin = (xx + n * (yy + n * zz)).astype(np.int32)
        _np(ids)[:] = np.where(border, lin, NONE).reshape(-1)

    @staticmethod
    def _dist(frame, ids, px, py, pz):
        n = frame.n
        f32 = np.float32
        vs, ox, oy, oz = f32(frame.voxel_size), f32(frame.origin[0]), f32(frame.origin[1]), f32(frame.origin[2])
        safe = np.where(ids == NONE, 0, ids)
        sx = ox + (safe % n).astype(f32) * vs
        sy = oy + ((safe // n) % n).astype(f32) * vs
        sz = oz + (safe // (n * n)).astype(f32) * vs
        return ((sx - px) * (sx - px) + (sy - py) * (sy - py)) + (sz - pz) * (sz - pz)

    def jfa_pass(self, frame, k, src, minus, plus, dst, algo):
        n, z0, z1 = frame.n, frame.z0, frame.z1
        nz = z1 - z0
        S = _np(src).reshape(nz, n, n)
        M = _np(minus).reshape(-1, n, n) if minus is not None else None
        P = _np(plus).reshape(-1, n, n) if plus is not None else None
        none_plane = np.full((n, n), NONE, np.int32)

        def plane(zg):                       # the addressing rule of vphip.h / vp_jfa_pass
            if zg < 0 or zg >= n:
                return none_plane
            if zg < z0:
                return M[zg - (z0 - k)]
            if zg >= z1:
                return P[zg - max(z1, z0 + k)]
            return S[zg - z0]

        _np(dst)[:] = self._pass(frame, k, plane).reshape(-1)

    # -- whole-grid buffers addressed by global plane (GhostSlabPipeline) --
    def jfa_pass_global(self, region, k, src_full, dst_full, algo):
        n = region.n
        A = _np(src_full).reshape(n, n, n)
        none_plane = np.full((n, n), NONE, np.int32)
        best = self._pass(region, k, lambda zg: A[zg] if 0 <= zg < n else none_plane)
        _np(dst_full).reshape(n, n, n)[region.z0:region.z1] = best

    def jfa_last_pass_global(self, region, src_full, scratch_full, words_region, fill, sdf, algo):
        n = region.n
        self.jfa_pass_global(region, 1, src_full, scratch_full, algo)
        ids = torch.from_numpy(_np(scratch_full).reshape(n, n, n)[region.z0:region.z1].reshape(-1).copy())
        self.jfa_finalize(region, words_region, ids, fill, sdf)

    # -- buffers that hold the planes [lo, hi) only (HybridSlabPipeline) --
    def jfa_pass_window(self, region, k, src, dst, lo, algo):
        n = region.n
        A = _np(src).reshape(-1, n, n)
        hi = lo + A.shape[0]
        none_plane = np.full((n, n), NONE, np.int32)

        def plane(zg):
            if zg < 0 or zg >= n:
                return none_plane
            assert lo <= zg < hi, "pass with step %d on [%d, %d) reads plane %d outside the window [%d, %d)" % (k, region.z0, region.z1, zg, lo, hi)
            return A[zg - lo]

        _np(dst).reshape(-1, n, n)[region.z0 - lo:region.z1 - lo] = self._pass(region, k, plane)

    def jfa_last_pass_window(self, region, src, scratch, lo, words_region, fill, sdf, algo):
        n = region.n
        self.jfa_pass_window(region, 1, src, scratch, lo, algo)
        ids = torch.from_numpy(_np(scratch).reshape(-1, n, n)[region.z0 - lo:region.z1 - lo].reshape(-1).copy())
        self.jfa_finalize(region, words_region, ids, fill, sdf)

    def _pass(self, frame, k, plane):
        n, z0, z1 = frame.n, frame.z0, frame.z1
        nz = z1 - z0
        f32 = np.float32
        vs, ox, oy, oz = f32(frame.voxel_size), f32(frame.origin[0]), f32(frame.origin[1]), f32(frame.origin[2])
        S = np.stack([plane(zg) for zg in range(z0, z1)], 0)
        zz, yy, xx = np.meshgrid(np.arange(z0, z1), np.arange(n), np.arange(n), indexing="ij")
        px = ox + xx.astype(f32) * vs
        py = oy + yy.astype(f32) * vs
        pz = oz + zz.astype(f32) * vs
        best = S.copy()
        bestd = np.where(best == NONE, f32(np.inf), self._dist(frame, best, px, py, pz)).astype(f32)
        for dz in (-1, 0, 1):
            stack = np.stack([plane(zg + dz * k) for zg in range(z0, z1)], 0)
            pad = np.pad(stack, ((0, 0), (k, k), (k, k)), constant_values=NONE)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx == 0 and dy == 0 and dz == 0:
                        continue
                    c = pad[:, k + dy * k:k + dy * k + n, k + dx * k:k + dx * k + n]
                    d = self._dist(frame, c, px, py, pz)
                    take = (c != NONE) & (d < bestd)
                    bestd = np.where(take, d, bestd)
                    best = np.where(take, c, best)
        return best

    def jfa_finalize(self, frame, words, ids, fill, sdf):
        n, nz = frame.n, frame.z1 - frame.z0
        f32 = np.float32
        vs, ox, oy, oz = f32(frame.voxel_size), f32(frame.origin[0]), f32(frame.origin[1]), f32(frame.origin[2])
        occ = self._bits(_np(words), n, nz)
        I = _np(ids).reshape(nz, n, n)
        zz, yy, xx = np.meshgrid(np.arange(frame.z0, frame.z1), np.arange(n), np.arange(n), indexing="ij")
        d = self._dist(frame, I, ox + xx.astype(f32) * vs, oy + yy.astype(f32) * vs, oz + zz.astype(f32) * vs)
        init = np.where(occ, f32(np.inf), f32(fill)).astype(f32)
        _np(sdf)[:] = np.where(I == NONE, init, np.copysign(d, init)).astype(f32).reshape(-1)
