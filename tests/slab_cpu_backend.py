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
    """The backend interface of slab.py (HipSlabBackend) on numpy: an id window is `planes` x n x n int32 linear voxel indices (NONE = no
    seed) -- the layout is the backend's own business, the pipelines only move whole planes of it (win_spans)."""

    tile_min_n = 0                            # no tile kernels here: every grid can be dealt cyclically (slab.cyclic_passes)

    def __init__(self, mesh_host, poison=None):
        self.mesh_host = mesh_host            # (xyz, tri) numpy: 'device' mesh handles are ignored
        self.poison = poison                  # tests: a voxel index every entry of a fresh window starts as: a (wrong) seed that would spoil the
                                              # result if a plane nobody produced were ever consumed

    def check_frame(self, frame):
        pass

    def empty_u32(self, n):
        return torch.zeros(int(n), dtype=torch.int32)

    def empty_f32(self, n):
        return torch.zeros(int(n), dtype=torch.float32)

    def can_start_from_mask(self, frame):
        return False

    # -- id windows ---------------------------------------------------------------------------
    def window(self, frame, planes):
        from cuda_mesh_voxelization_amd.slab import IdWindow
        return IdWindow(torch.full((int(planes) * frame.n * frame.n,), int(NONE if self.poison is None else self.poison), dtype=torch.int32), planes)

    def win_spans(self, frame, w, p0, p1):
        pv = frame.n * frame.n
        assert 0 <= p0 <= p1 <= w.planes, "planes [%d, %d) outside a window of %d" % (p0, p1, w.planes)
        return [w.t[p0 * pv:p1 * pv]]

    def _planes(self, frame, w, p0, p1):
        return _np(self.win_spans(frame, w, p0, p1)[0]).reshape(p1 - p0, frame.n, frame.n)

    def win_init(self, region, words_region, below, above, w, at):
        nz = region.z1 - region.z0
        ids = torch.zeros(nz * region.n * region.n, dtype=torch.int32)
        self.jfa_init(region, words_region, below, above, ids)
        self._planes(region, w, at, at + nz)[:] = _np(ids).reshape(nz, region.n, region.n)

    def surface(self, frame, words, border):
        ids = torch.zeros(frame.voxels, dtype=torch.int32)
        self.jfa_init(frame, words, None, None, ids)
        bits = (_np(ids) != NONE)
        _np(border)[:] = np.packbits(bits, bitorder="little").view(np.int32)

    def win_first_two(self, frame, border_full, w):
        """passes n/2 and n/4 of the whole grid from its border mask"""
        from cuda_mesh_voxelization_amd.slab import IdWindow
        n = frame.n
        assert w.planes == n
        bits = np.unpackbits(_np(border_full).view(np.uint8), bitorder="little").astype(bool)
        lin = np.arange(n * n * n, dtype=np.int32)
        tmp = [IdWindow(torch.from_numpy(np.where(bits, lin, NONE).astype(np.int32)), n), IdWindow(torch.zeros(n * n * n, dtype=torch.int32), n)]
        self.win_pass(frame, n // 2, tmp[0], tmp[1], 0)
        self.win_pass(frame, n // 4, tmp[1], w, 0)

    def win_pass(self, region, k, w_in, w_out, at, stride=None):
        n, z0, z1 = region.n, region.z0, region.z1
        stride = k if stride is None else stride
        assert w_in.planes == w_out.planes
        A = _np(w_in.t).reshape(w_in.planes, n, n)
        none_plane = np.full((n, n), NONE, np.int32)

        def plane(z, d):                      # the plane z + d k of a region plane z sits d * stride planes from it in the window
            if z + d * k < 0 or z + d * k >= n:
                return none_plane
            i = at + (z - z0) + d * stride
            assert 0 <= i < w_in.planes, "pass with step %d (stride %d) on [%d, %d) at %d reads window plane %d of %d" % (k, stride, z0, z1, at, i, w_in.planes)
            return A[i]

        _np(w_out.t).reshape(w_out.planes, n, n)[at:at + (z1 - z0)] = self._pass(region, k, plane)

    # -- cyclic plane distribution (TransposeSlabPipeline): local plane l of rank r = global plane r + l * world ----------
    def cyclic_passes(self, frame, world):
        from cuda_mesh_voxelization_amd.slab import cyclic_passes
        return cyclic_passes(frame.n, world, self.tile_min_n)

    def win_first_two_cyclic(self, frame, border_full, w, world, rank):
        """passes n/2 and n/4 of the rank's planes (z = rank mod world) from the whole-grid border mask"""
        from cuda_mesh_voxelization_amd.slab import IdWindow
        n = frame.n
        assert w.planes == n // world and (n // 4) % world == 0
        bits = np.unpackbits(_np(border_full).view(np.uint8), bitorder="little").astype(bool).reshape(n, n, n)
        lin = np.arange(n * n * n, dtype=np.int32).reshape(n, n, n)
        mine = np.where(bits, lin, NONE).astype(np.int32)[rank::world]
        tmp = [IdWindow(torch.from_numpy(mine.reshape(-1).copy()), w.planes), IdWindow(torch.zeros(w.planes * n * n, dtype=torch.int32), w.planes)]
        self.win_pass_cyclic(frame, n // 2, tmp[0], tmp[1], world, rank)
        self.win_pass_cyclic(frame, n // 4, tmp[1], w, world, rank)

    def win_pass_cyclic(self, frame, k, w_in, w_out, world, rank):
        n = frame.n
        assert k % world == 0 and w_in.planes == w_out.planes == n // world
        A = _np(w_in.t).reshape(w_in.planes, n, n)
        none_plane = np.full((n, n), NONE, np.int32)
        zs = [rank + world * l for l in range(w_in.planes)]

        def plane(z, d):
            zz = z + d * k
            if zz < 0 or zz >= n:
                return none_plane
            assert (zz - rank) % world == 0, "cyclic pass with step %d reads plane %d, which rank %d of %d does not hold" % (k, zz, rank, world)
            return A[(zz - rank) // world]

        _np(w_out.t).reshape(w_out.planes, n, n)[:] = self._pass(frame, k, plane, zs)

    def win_interleave(self, frame, w_in, w_out, at, world, count):
        n = frame.n
        assert w_in.planes == world * count and 0 <= at and at + world * count <= w_out.planes
        src = _np(w_in.t).reshape(world, count, n, n)
        _np(w_out.t).reshape(w_out.planes, n, n)[at:at + world * count] = src.transpose(1, 0, 2, 3).reshape(world * count, n, n)

    def win_last_pass(self, region, w_in, w_scratch, at, words_region, fill, sdf, stride=1):
        n, nz = region.n, region.z1 - region.z0
        self.win_pass(region, 1, w_in, w_scratch, at, stride)
        ids = torch.from_numpy(_np(w_scratch.t).reshape(-1, n, n)[at:at + nz].reshape(-1).copy())
        self.jfa_finalize(region, words_region, ids, fill, sdf)

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

    def _pass(self, frame, k, plane, zs=None):
        """one pass with step k over the output planes zs (global plane numbers; default: the planes of the frame); plane(z, d) = the id
        plane z + d k as the caller's window holds it"""
        n = frame.n
        zs = list(range(frame.z0, frame.z1)) if zs is None else list(zs)
        f32 = np.float32
        vs, ox, oy, oz = f32(frame.voxel_size), f32(frame.origin[0]), f32(frame.origin[1]), f32(frame.origin[2])
        S = np.stack([plane(zg, 0) for zg in zs], 0)
        zz, yy, xx = np.meshgrid(np.array(zs), np.arange(n), np.arange(n), indexing="ij")
        px = ox + xx.astype(f32) * vs
        py = oy + yy.astype(f32) * vs
        pz = oz + zz.astype(f32) * vs
        best = S.copy()
        bestd = np.where(best == NONE, f32(np.inf), self._dist(frame, best, px, py, pz)).astype(f32)
        for dz in (-1, 0, 1):
            stack = np.stack([plane(zg, dz) for zg in zs], 0)
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
