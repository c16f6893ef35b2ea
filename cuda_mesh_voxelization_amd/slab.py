"""Z-slab sharding of one n^3 job over the GPUs of a node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) point-to-point halo exchange between JFA steps.

The reference is single-GPU; this is the multi-GPU extension the north star
asks for.  Rank r owns the planes [r*nz, (r+1)*nz), nz = n / world.

  voxelize   every (y,z) column is independent (vox/sequential.cpp:40-57): each rank rasterises the
             whole (small) mesh into its own slab -- no exchange.
  CSG        word-wise -- no exchange.
  JFA init   26-neighbourhood -> one bitmask plane from each Z-neighbour.
  JFA pass k voxel z reads planes z-k, z, z+k -> before the pass every rank receives the id planes
             [z0-k, min(z0, z1-k)) and [max(z1, z0+k), z1+k) from whoever owns them (for k >= nz
             these are whole slabs of ranks r -+ k/nz), as one batch of isend/irecv.
  finalize   local.

Every stage is a pure function of the previous buffers, so the concatenated slabs are bit-identical
to the single-GPU result for any world size -- that is the parity test (tests/test_slab_*.py).

The compute backend is pluggable only so that the exchange logic can be exercised on CPU with gloo
in the test-suite; the product backend (HipSlabBackend) calls libvphip.so and nothing else.
"""
from __future__ import annotations

import math

import torch

from .capi import ALGO_TILED, Frame


class HipSlabBackend:
    """Product backend: torch CUDA tensors for memory, libvphip.so (via Engine.ctx) for compute."""

    def __init__(self, engine):
        self.engine = engine
        self.ctx = engine.ctx
        self.device = engine.device

    def empty_u32(self, n):
        return torch.empty(int(n), dtype=torch.int32, device=self.device)

    def empty_f32(self, n):
        return torch.empty(int(n), dtype=torch.float32, device=self.device)

    def ids_u32(self, n):
        """An id volume of the ghost / hybrid pipelines: filled ONCE at allocation.  Their regions are rounded outwards to the
        8-plane tile, and the excess planes of a pass read planes the pass before it never produced (ghost_regions): what they
        read is then this fill or stale ids -- never memory nobody wrote.  VP_SLAB_POISON=<byte> (tests) fills with that byte
        instead of 0 to show that the results do not depend on it."""
        import os
        t = torch.empty(int(n), dtype=torch.int32, device=self.device)
        t.view(torch.uint8).fill_(int(os.environ.get("VP_SLAB_POISON", "0"), 0) & 0xFF)
        return t

    def id_words(self, frame):
        """uint32 words of JFA state per voxel (1 for n <= 1024, 2 above)."""
        return self.ctx.jfa_id_bytes(frame) // 4

    @staticmethod
    def _p(t):
        return t.data_ptr() if t is not None else None

    def voxelize(self, frame, words, d_xyz, d_tri, algo):
        self.ctx.voxelize(frame, words.data_ptr(), d_xyz.data_ptr(), d_xyz.shape[0], d_tri.data_ptr(), d_tri.shape[0], algo, False)

    def csg(self, a, b, op):
        self.ctx.csg(a.data_ptr(), b.data_ptr(), a.numel(), op)

    def jfa_init(self, frame, words, below, above, ids):
        self.ctx.jfa_init(frame, words.data_ptr(), self._p(below), self._p(above), ids.data_ptr())

    def jfa_pass(self, frame, k, src, minus, plus, dst, algo):
        self.ctx.jfa_pass(frame, k, src.data_ptr(), self._p(minus), self._p(plus), dst.data_ptr(), algo)

    def jfa_finalize(self, frame, words, ids, fill, sdf):
        self.ctx.jfa_finalize(frame, words.data_ptr(), ids.data_ptr(), fill, sdf.data_ptr())

    def jfa_last_pass(self, frame, src, minus, plus, scratch, words, fill, sdf, algo):
        self.ctx.jfa_last_pass(frame, src.data_ptr(), self._p(minus), self._p(plus), scratch.data_ptr(), words.data_ptr(),
                               fill, sdf.data_ptr(), algo)

    # -- whole-grid id buffers addressed by global plane (GhostSlabPipeline) --------------------
    def _global_ptrs(self, region, k, src_full, dst_full):
        pb = region.n * region.n * self.ctx.jfa_id_bytes(region)    # bytes per id plane
        s, d = src_full.data_ptr(), dst_full.data_ptr()
        # vphip.h / vp_jfa_pass: plane p of d_minus is global plane z0-k+p, d_plus starts at max(z1, z0+k)
        return (s + region.z0 * pb, s + (region.z0 - k) * pb, s + max(region.z1, region.z0 + k) * pb, d + region.z0 * pb)

    def jfa_pass_global(self, region, k, src_full, dst_full, algo):
        src, minus, plus, dst = self._global_ptrs(region, k, src_full, dst_full)
        self.ctx.jfa_pass(region, k, src, minus, plus, dst, algo)

    # -- id buffers that hold the planes [lo, hi) only, addressed by global plane (HybridSlabPipeline) --------------
    def _window_ptrs(self, region, k, src, dst, lo):
        pb = region.n * region.n * self.ctx.jfa_id_bytes(region)
        s, d = src.data_ptr() - lo * pb, dst.data_ptr() - lo * pb      # where plane 0 would be
        return (s + region.z0 * pb, s + (region.z0 - k) * pb, s + max(region.z1, region.z0 + k) * pb, d + region.z0 * pb)

    def jfa_pass_window(self, region, k, src, dst, lo, algo):
        a, minus, plus, out = self._window_ptrs(region, k, src, dst, lo)
        self.ctx.jfa_pass(region, k, a, minus, plus, out, algo)

    def jfa_last_pass_window(self, region, src, scratch, lo, words_region, fill, sdf, algo):
        a, minus, plus, out = self._window_ptrs(region, 1, src, scratch, lo)
        self.ctx.jfa_last_pass(region, a, minus, plus, out, words_region.data_ptr(), fill, sdf.data_ptr(), algo)

    def jfa_first_pass_window(self, region, border_full, dst, lo):
        pb = region.n * region.n * self.ctx.jfa_id_bytes(region)
        self.ctx.jfa_first_pass(region, border_full.data_ptr(), dst.data_ptr() + (region.z0 - lo) * pb)

    def can_start_from_mask(self, frame, algo):
        return self.ctx.jfa_can_start_from_mask(frame, algo)

    def surface(self, frame, words, border):
        self.ctx.surface(frame, words.data_ptr(), None, None, border.data_ptr())

    def can_fuse_first_two(self, frame, algo):
        return self.ctx.jfa_can_fuse_first_two(frame, algo)

    def jfa_first_two_global(self, frame, border_full, dst_full):
        self.ctx.jfa_first_two(frame, border_full.data_ptr(), dst_full.data_ptr())

    def jfa_first_pass_global(self, region, border_full, dst_full):
        self.ctx.jfa_first_pass(region, border_full.data_ptr(),
                                dst_full.data_ptr() + region.z0 * region.n * region.n * self.ctx.jfa_id_bytes(region))

    def jfa_last_pass_global(self, region, src_full, scratch_full, words_region, fill, sdf, algo):
        src, minus, plus, scratch = self._global_ptrs(region, 1, src_full, scratch_full)
        self.ctx.jfa_last_pass(region, src, minus, plus, scratch, words_region.data_ptr(), fill, sdf.data_ptr(), algo)

    # -- whole-volume calls (vp_jfa_volume_*): volume base + region frame, the library picks the layout (compact above n = 1024)
    def volume_words(self, frame):
        """uint32 words of one id volume of the whole grid in the library's layout"""
        return (self.ctx.jfa_volume_bytes(frame) + 3) // 4

    def jfa_volume_first_two(self, frame, border_full, vol):
        self.ctx.jfa_volume_first_two(frame, border_full.data_ptr(), vol.data_ptr())

    def jfa_volume_pass(self, region, k, vol_in, vol_out):
        self.ctx.jfa_volume_pass(region, k, vol_in.data_ptr(), vol_out.data_ptr())

    def jfa_volume_last_pass(self, region, vol_in, vol_scratch, words_region, fill, sdf):
        self.ctx.jfa_volume_last_pass(region, vol_in.data_ptr(), vol_scratch.data_ptr(), words_region.data_ptr(), fill, sdf.data_ptr())


class HostStagedDist:
    """Test rigs only (bench.py with VP_BENCH_SHARE_GPU=1: ranks share a GPU and rendezvous over gloo, whose send / recv take CPU
    tensors): the point-to-point subset the pipelines use, staged through host memory.  On a real node the pipelines get
    torch.distributed itself (backend "nccl" = RCCL) and device tensors travel directly."""
    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Req:
        def __init__(self, reqs, tensor=None, host=None):
            self.reqs, self.tensor, self.host = reqs, tensor, host

        def wait(self):
            for r in self.reqs:
                r.wait()
            if self.tensor is not None:
                self.tensor.copy_(self.host)

    def __init__(self, dist):
        self.dist = dist

    def batch_isend_irecv(self, ops):
        out = []
        for o in ops:
            if o.op == "isend":
                host = o.tensor.cpu()
                out.append(self._Req([self.dist.isend(host, o.peer)], None, host))
        for o in ops:
            if o.op == "irecv":
                host = torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                out.append(self._Req([self.dist.irecv(host, o.peer)], o.tensor, host))
        return out

    def barrier(self):
        self.dist.barrier()


def hbm_bytes(obj) -> int:
    """bytes of device memory held by the tensors of a pipeline object (attributes, and lists / dicts of them), each storage once"""
    seen, total = set(), 0

    def walk(v):
        nonlocal total
        if isinstance(v, torch.Tensor):
            st = v.untyped_storage()
            if st.data_ptr() not in seen:
                seen.add(st.data_ptr())
                total += st.nbytes()
        elif isinstance(v, (list, tuple)):
            for x in v:
                walk(x)
        elif isinstance(v, dict):
            for x in v.values():
                walk(x)

    for v in vars(obj).values():
        walk(v)
    return total


def slab_range(n: int, rank: int, world: int):
    if n % world != 0 or (n // world) % 8 != 0:
        raise ValueError("n=%d cannot be cut into %d Z-slabs of a multiple of 8 planes" % (n, world))
    nz = n // world
    return rank * nz, (rank + 1) * nz


def halo_plan(n: int, world: int, k: int):
    """For step k: list of (src_rank, dst_rank, side, g0, g1) = dst needs global planes [g0, g1) owned
    by src, for its 'minus' or 'plus' buffer.  Deterministic and identical on every rank."""
    nz = n // world
    plan = []
    for dst in range(world):
        z0, z1 = dst * nz, (dst + 1) * nz
        regions = (("minus", max(z0 - k, 0), min(z0, z1 - k)), ("plus", max(z1, z0 + k), min(z1 + k, n)))
        for side, a, b in regions:
            g = a
            while g < b:
                src = g // nz
                e = min(b, (src + 1) * nz)
                plan.append((src, dst, side, g, e))
                g = e
    return plan


class SlabPipeline:
    """voxelize -> (CSG) -> JFA for the slab of this rank."""

    def __init__(self, backend, frame: Frame, rank: int, world: int, dist):
        self.be = backend
        self.dist = dist
        self.rank, self.world = rank, world
        self.global_frame = frame
        self.z0, self.z1 = slab_range(frame.n, rank, world)
        self.frame = frame.slab(self.z0, self.z1)
        n = frame.n
        idw = self.be.id_words(frame) if hasattr(self.be, "id_words") else 1
        self.plane_ids = n * n * idw              # uint32 words of id state per plane
        self.plane_words = n * n // 32            # bitmask words per plane
        self.nz = self.z1 - self.z0
        self.words = self.be.empty_u32(self.frame.words)
        # Id volumes with room for the halo planes of the narrow passes (k <= nz/2) directly below and above the slab:
        # [H | nz | H] planes, H = nz/2.  For those passes the received planes land next to the slab, the three buffers
        # vp_jfa_pass takes are ONE contiguous volume, and the dense tile kernel (jfa.hip: jfa_pass_dense) applies.
        # Wide passes (k >= nz) receive whole slabs of distant ranks into the separate minus / plus buffers.
        self.H = (self.nz // 2) if world > 1 else 0
        self._bufs = [self.be.empty_u32((self.nz + 2 * self.H) * self.plane_ids) for _ in range(2)]
        self.ids = [b[self.H * self.plane_ids:(self.H + self.nz) * self.plane_ids] for b in self._bufs]
        self.sdf = self.be.empty_f32(self.frame.voxels)
        self.minus = self.be.empty_u32(self.frame.voxels * idw) if world > 1 else None
        self.plus = self.be.empty_u32(self.frame.voxels * idw) if world > 1 else None
        self.below = self.be.empty_u32(self.plane_words) if rank > 0 else None
        self.above = self.be.empty_u32(self.plane_words) if rank < world - 1 else None
        self.bytes_received = 0

    def describe(self):
        return "z-slab x%d, RCCL p2p halo exchange before every pass" % self.world

    def report(self):
        return {"pipeline": "halo", "slab_planes": self.nz, "bytes_received_total": int(self.bytes_received),
                "hbm_bytes_this_rank": hbm_bytes(self)}

    # -- stages ---------------------------------------------------------------------------
    def voxelize(self, d_xyz, d_tri, algo=ALGO_TILED, out=None):
        out = self.words if out is None else out
        self.be.voxelize(self.frame, out, d_xyz, d_tri, algo)
        return out

    def csg(self, other, op: int):
        self.be.csg(self.words, other, op)
        return self.words

    def _exchange(self, ops):
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()

    def _exchange_mask_planes(self):
        d, P = self.dist, self.dist.P2POp
        ops = []
        pw = self.plane_words
        if self.rank > 0:                                         # my first plane is the 'above' plane of rank-1
            ops.append(P(d.isend, self.words[:pw], self.rank - 1))
            ops.append(P(d.irecv, self.below, self.rank - 1))
        if self.rank < self.world - 1:                            # my last plane is the 'below' plane of rank+1
            ops.append(P(d.isend, self.words[(self.nz - 1) * pw:], self.rank + 1))
            ops.append(P(d.irecv, self.above, self.rank + 1))
        self._exchange(ops)

    def _halo_buffers(self, k: int, src):
        """(minus, plus) for step k around the id volume `src` (one of self.ids): views into src's own allocation when
        the halo fits next to the slab (k <= H), else the separate whole-slab buffers."""
        if self.world == 1:
            return None, None
        if k <= self.H:
            buf = self._bufs[0] if src.data_ptr() == self.ids[0].data_ptr() else self._bufs[1]
            pi = self.plane_ids
            return buf[(self.H - k) * pi:self.H * pi], buf[(self.H + self.nz) * pi:(self.H + self.nz + k) * pi]
        return self.minus, self.plus

    def _exchange_ids(self, k: int, src):
        d, P = self.dist, self.dist.P2POp
        pi = self.plane_ids
        minus_base = self.z0 - k                                  # global plane of minus[0] (vphip.h, vp_jfa_pass)
        plus_base = max(self.z1, self.z0 + k)
        minus, plus = self._halo_buffers(k, src)
        ops = []
        for s, t, side, g0, g1 in halo_plan(self.global_frame.n, self.world, k):
            if s == t:
                continue
            if s == self.rank:
                ops.append(P(d.isend, src[(g0 - self.z0) * pi:(g1 - self.z0) * pi], t))
            elif t == self.rank:
                buf, base = (minus, minus_base) if side == "minus" else (plus, plus_base)
                ops.append(P(d.irecv, buf[(g0 - base) * pi:(g1 - base) * pi], s))
                self.bytes_received += (g1 - g0) * pi * 4
        self._exchange(ops)

    def jfa(self, algo=ALGO_TILED, fill=-math.inf, out=None):
        out = self.sdf if out is None else out
        if self.world > 1:
            self._exchange_mask_planes()
        a, b = self.ids
        self.be.jfa_init(self.frame, self.words, self.below, self.above, a)
        k = self.global_frame.n // 2
        while k >= 1:                                             # jfa/sequential.cpp:72
            if self.world > 1:
                self._exchange_ids(k, a)
            minus, plus = self._halo_buffers(k, a)
            if k == 1 and hasattr(self.be, "jfa_last_pass"):       # last pass + finalize fused
                self.be.jfa_last_pass(self.frame, a, minus, plus, b, self.words, fill, out, algo)
                return out
            self.be.jfa_pass(self.frame, k, a, minus, plus, b, algo)
            a, b = b, a
            k //= 2
        self.be.jfa_finalize(self.frame, self.words, a, fill, out)
        return out


# =============================================================================================
# Communication-avoiding variant ("ghost zones")
# =============================================================================================
def ghost_regions(n: int, rank: int, world: int):
    """Planes [b0, b1) each JFA pass must produce on this rank so that NO exchange is needed:
    the pass with step k_i feeds every later pass, so it has to cover the slab widened by the sum of
    the later steps (= k_i - 1 for a halving sequence, its REACH g_i), rounded outwards to the 8-plane
    tile and clipped to the grid.  Returns [(k, b0, b1), ...] in pass order.

    Invariant (tests: test_ghost_ignores_unproduced_planes, CPU and GPU): a plane of region i is NEEDED iff it
    lies within g_i of the slab; needed planes of pass i read only planes within g_i + k_i = g_(i-1) of
    the slab, all of which pass i - 1 produced.  The planes the rounding adds are computed as well (whole
    tiles), from planes pass i - 1 may NOT have produced; no needed plane of a later pass ever reads them,
    so the slab is exact whatever they held (the id volumes are filled once at allocation, ids_u32, so
    that it is never unwritten memory).  Nesting the regions instead (each containing the next one widened
    by its step) would cost up to 16 more planes per side and pass for values nobody reads."""
    z0, z1 = slab_range(n, rank, world)
    ks = []
    k = n // 2
    while k >= 1:
        ks.append(k)
        k //= 2
    out = []
    for i, k in enumerate(ks):
        g = sum(ks[i + 1:])
        b0 = max(0, (z0 - g) // 8 * 8)
        b1 = min(n, -((-(z1 + g)) // 8) * 8)
        out.append((k, b0, b1))
    return out


def fused_first_two_threshold(n: int) -> int:
    """Per cent of the grid the second pass of a rank must cover before the whole-grid launch of the first two passes is the cheaper
    way to run them.  Break-even of the measured kernel times: jfa_first_two 0.30 ms at n = 512 against 0.18 for the first pass + 0.36 x
    the covered fraction for the second (33 %); 2.13 ms against 1.39 + 2.12 x the fraction at n = 1024 (35 %).  Swept on the GPU
    (profiles/r03/slab_scaling_*.txt: 35 against round 2's 65 / 73): 8 slabs 1.26 -> 1.09 ms at n = 512, 11.2 -> 9.1 ms at n = 1024,
    195 -> 144 ms at n = 2048.  The same rule lives in csrc/multi.hip."""
    import os
    if os.environ.get("VP_FUSED_FIRST_TWO_PCT"):                   # dev: tools/slab_scaling.py sweeps it
        return int(os.environ["VP_FUSED_FIRST_TWO_PCT"])
    return 35


class GhostSlabPipeline:
    """Z-slab strong scaling WITHOUT halo exchange.

    Measured on MI355X one JFA pass costs ~1.6 us per 512^2 plane, while moving that plane (1 MiB) to a
    peer over one xGMI link costs ~20 us: recomputing ghost planes is an order of magnitude cheaper than
    exchanging them.  Every rank therefore voxelizes and initialises the whole grid (0.2 ms at n = 512)
    and runs pass i on its slab widened by the reach of the later passes (ghost_regions).  The regions
    shrink to the bare slab at k = 1; the result is the slab's part of the single-GPU result, bit for
    bit, with zero bytes exchanged.  Id buffers are addressed by global plane, so each rank holds two
    full id volumes (2 x 4 n^3 B: 1 GiB at n = 512, 8 GiB at n = 1024 -- small against 288 GB).
    SlabPipeline above (RCCL point-to-point halos) remains for grids whose state does not fit.
    """

    def __init__(self, backend, frame: Frame, rank: int, world: int):
        self.be = backend
        self.rank, self.world = rank, world
        self.global_frame = frame
        self.z0, self.z1 = slab_range(frame.n, rank, world)
        self.frame = frame.slab(self.z0, self.z1)
        self.regions = ghost_regions(frame.n, rank, world)
        self.words = self.be.empty_u32(frame.words)                 # whole grid
        idw = self.be.id_words(frame) if hasattr(self.be, "id_words") else 1
        # Whole-volume calls (round 4) where the first two passes are fused over the whole grid anyway -- always above n = 1024, where the
        # library then keeps the volumes in its compact 5-byte layout: 2 x 40 instead of 2 x 64 GiB at n = 2048 and 10 instead of 16 bytes
        # per voxel and pass.  VP_GHOST_VOLUME=0 (dev / tests) keeps the caller-addressed 8-byte planes.
        import os
        self.volume_mode = (hasattr(self.be, "jfa_volume_pass") and os.environ.get("VP_GHOST_VOLUME", "1") != "0" and len(self.regions) > 3
                            and self.be.can_fuse_first_two(frame, ALGO_TILED)
                            and (frame.n > 1024 or (self.regions[1][2] - self.regions[1][1]) * 100 >= fused_first_two_threshold(frame.n) * frame.n))
        vol_words = self.be.volume_words(frame) if self.volume_mode else frame.voxels * idw
        self.ids = [(self.be.ids_u32(vol_words) if hasattr(self.be, "ids_u32") else self.be.empty_u32(vol_words)) for _ in range(2)]
        self.sdf = self.be.empty_f32(self.frame.voxels)             # own slab only
        self.planes_computed = sum(b1 - b0 for _, b0, b1 in self.regions)
        self.border = None

    def describe(self):
        return "z-slab x%d, ghost planes recomputed, no data-path exchange" % self.world

    def report(self):
        n, passes = self.global_frame.n, len(self.regions)
        return {"pipeline": "ghost", "slab_planes": self.z1 - self.z0, "regions": [[k, b0, b1] for k, b0, b1 in self.regions],
                "plane_passes_this_rank": int(self.planes_computed), "plane_passes_one_gpu": n * passes,
                "plane_pass_ratio": round(n * passes / self.planes_computed, 3), "bytes_exchanged": 0, "bytes_received_total": 0,
                "first_two_passes_fused_over_whole_grid": bool(getattr(self, "fused_first_two", False)),
                "volume_calls": bool(self.volume_mode), "id_volume_bytes": int(self.ids[0].numel() * 4),
                "hbm_bytes_this_rank": hbm_bytes(self)}

    def voxelize(self, d_xyz, d_tri, algo=ALGO_TILED, out=None):
        out = self.words if out is None else out
        self.be.voxelize(self.global_frame, out, d_xyz, d_tri, algo)
        return out

    def csg(self, other, op: int):
        self.be.csg(self.words, other, op)
        return self.words

    def _jfa_volume(self, fill, out):
        """the whole-volume sequence: border mask, passes n/2 + n/4 over the whole grid in one launch, every later pass on its region"""
        a, b = self.ids
        G = self.global_frame
        if self.border is None:
            self.border = self.be.empty_u32(G.words)
        self.be.surface(G, self.words, self.border)
        self.be.jfa_volume_first_two(G, self.border, a)
        self.fused_first_two = True
        pw = G.n * G.n // 32
        last = len(self.regions) - 1
        for i, (k, b0, b1) in enumerate(self.regions):
            if i < 2:
                continue
            region = G.slab(b0, b1)
            if i == last:
                self.be.jfa_volume_last_pass(region, a, b, self.words[b0 * pw:b1 * pw], fill, out)
                return out
            self.be.jfa_volume_pass(region, k, a, b)
            a, b = b, a
        return out

    def jfa(self, algo=ALGO_TILED, fill=-math.inf, out=None):
        out = self.sdf if out is None else out
        if self.volume_mode and algo == ALGO_TILED:
            return self._jfa_volume(fill, out)
        need = self.global_frame.voxels * (self.be.id_words(self.global_frame) if hasattr(self.be, "id_words") else 1)
        if self.ids[0].numel() < need:                            # the volumes were sized for the library's layout: caller-addressed ids need more
            self.ids = [(self.be.ids_u32(need) if hasattr(self.be, "ids_u32") else self.be.empty_u32(need)) for _ in range(2)]
        a, b = self.ids
        last = len(self.regions) - 1
        mask_start = last > 0 and hasattr(self.be, "can_start_from_mask") and self.be.can_start_from_mask(self.global_frame, algo)
        if mask_start:
            # border mask of the whole grid -> first pass directly (no init id volume), see vp_jfa_first_pass
            if self.border is None:
                self.border = self.be.empty_u32(self.global_frame.words)
            self.be.surface(self.global_frame, self.words, self.border)
        else:
            self.be.jfa_init(self.global_frame, self.words, None, None, a)
        # The first two passes (k = n/2, n/4) as ONE launch over the whole grid (vp_jfa_first_two) where the second pass would cover
        # most of the grid anyway: 0.40 ms at n = 512 against 0.18 for the first pass + 0.36 x the covered fraction for the second
        # (break-even at 61 % in round 2; the kernel has since come down to 0.30 / 2.13 ms: 35 %, see fused_first_two_threshold()).
        skip = 0
        if (mask_start and last >= 2 and hasattr(self.be, "can_fuse_first_two") and self.be.can_fuse_first_two(self.global_frame, algo)
                and (self.regions[1][2] - self.regions[1][1]) * 100 >= fused_first_two_threshold(self.global_frame.n) * self.global_frame.n):
            self.be.jfa_first_two_global(self.global_frame, self.border, b)
            a, b = b, a
            skip = 2
            self.fused_first_two = True
        for i, (k, b0, b1) in enumerate(self.regions):
            if i < skip:
                continue
            region = self.global_frame.slab(b0, b1)
            if i == 0 and mask_start:
                self.be.jfa_first_pass_global(region, self.border, b)
                a, b = b, a
                continue
            if i == last:
                pw = self.global_frame.n * self.global_frame.n // 32
                self.be.jfa_last_pass_global(region, a, b, self.words[b0 * pw:b1 * pw], fill, out, algo)
                return out
            self.be.jfa_pass_global(region, k, a, b, algo)
            a, b = b, a
        return out


# =============================================================================================
# Hybrid: ghost planes where planes are cheap to recompute and dear to move, halos where it is the other way round
# =============================================================================================
def hybrid_plan(n: int, rank: int, world: int):
    """(wide, narrow): the passes with k > nz/2 as [(k, b0, b1)] -- run on the slab widened by the reach of the LATER WIDE passes
    only (rounded to 8 planes, clipped) -- and the steps k <= nz/2, which run on the bare slab behind a halo of k planes from each
    adjacent rank.  world == 1: every pass is 'wide' with the whole grid as its region."""
    z0, z1 = slab_range(n, rank, world)
    H = (z1 - z0) // 2 if world > 1 else 0
    ks = []
    k = n // 2
    while k >= 1:
        ks.append(k)
        k //= 2
    wide_ks = [k for k in ks if k > H]
    wide = []
    for i, k in enumerate(wide_ks):
        g = sum(wide_ks[i + 1:])
        wide.append((k, max(0, (z0 - g) // 8 * 8), min(n, -((-(z1 + g)) // 8) * 8)))
    return wide, [k for k in ks if k <= H]


class HybridSlabPipeline:
    """Z-slabs, the two ways of feeding a pass mixed by what each costs.

    A pass with step k needs the planes z +- k.  For the WIDE passes (k > nz/2: whole slabs of distant ranks) the planes are
    recomputed as ghost planes, as in GhostSlabPipeline -- but only as far as the later WIDE passes reach, because the NARROW
    passes (k <= nz/2) fetch their k halo planes from the two adjacent ranks (point-to-point, one send and one receive per side
    and pass).  The halos of the NEXT pass are sent as soon as the boundary planes of the current one are computed -- those
    sub-slabs are launched first -- so the transfer runs under the interior planes of the current pass.

    Plane-passes per rank: sum of the wide regions + nz per narrow pass (n = 1024, 8 ranks: 2,304 against 3,220 for ghost planes
    alone and 1,280 ideal); received: sum of the narrow k = nz - 1 planes per side and job.  Id buffers hold the planes of the
    largest wide region (not the grid); the narrow passes run in place inside them.
    """

    def __init__(self, backend, frame: Frame, rank: int, world: int, dist):
        self.be, self.dist = backend, dist
        self.rank, self.world = rank, world
        self.global_frame = frame
        self.z0, self.z1 = slab_range(frame.n, rank, world)
        self.nz = self.z1 - self.z0
        self.frame = frame.slab(self.z0, self.z1)
        self.wide, self.narrow = hybrid_plan(frame.n, rank, world)
        idw = self.be.id_words(frame) if hasattr(self.be, "id_words") else 1
        self.plane_ids = frame.n * frame.n * idw
        self.plane_words = frame.n * frame.n // 32
        self.words = self.be.empty_u32(frame.words)                  # whole grid: every rank rasterises it (0.06 ms at n = 512)
        self.sdf = self.be.empty_f32(self.frame.voxels)
        self.border = None
        self._bufs = {}
        self.bytes_received = 0
        self.planes_computed = sum(b1 - b0 for _, b0, b1 in self.wide) + self.nz * len(self.narrow)

    def describe(self):
        return "z-slab x%d, hybrid: ghost planes for k > nz/2, p2p halos under the interior planes for k <= nz/2" % self.world

    def report(self):
        n, passes = self.global_frame.n, len(self.wide) + len(self.narrow)
        return {"pipeline": "hybrid", "slab_planes": self.nz, "wide_regions": [[k, b0, b1] for k, b0, b1 in self.wide],
                "narrow_steps": list(self.narrow), "plane_passes_this_rank": int(self.planes_computed),
                "plane_passes_one_gpu": n * passes, "plane_pass_ratio": round(n * passes / self.planes_computed, 3),
                "bytes_received_total": int(self.bytes_received), "id_buffer_planes": getattr(self, "window", None),
                "hbm_bytes_this_rank": hbm_bytes(self)}

    def voxelize(self, d_xyz, d_tri, algo=ALGO_TILED, out=None):
        out = self.words if out is None else out
        self.be.voxelize(self.global_frame, out, d_xyz, d_tri, algo)
        return out

    def csg(self, other, op: int):
        self.be.csg(self.words, other, op)
        return self.words

    # -- buffers: planes [lo, hi) of the id volume, twice ------------------------------------
    def _window(self, mask_start: bool):
        n, H = self.global_frame.n, (self.nz // 2 if self.world > 1 else 0)
        lo, hi = max(0, self.z0 - H), min(n, self.z1 + H)            # room for the halos of the narrow passes
        for i, (k, b0, b1) in enumerate(self.wide):
            lo, hi = min(lo, b0), max(hi, b1)
            if i > 0 or not mask_start:                               # every pass that reads ids reads k planes beyond its region
                lo, hi = min(lo, max(0, b0 - k)), max(hi, min(n, b1 + k))
        key = (lo, hi)
        if key not in self._bufs:
            self._bufs = {key: [(self.be.ids_u32((hi - lo) * self.plane_ids) if hasattr(self.be, "ids_u32") else self.be.empty_u32((hi - lo) * self.plane_ids)) for _ in range(2)]}
        self.window = [lo, hi]
        return lo, hi, self._bufs[key]

    def _planes(self, buf, lo, g0, g1):
        return buf[(g0 - lo) * self.plane_ids:(g1 - lo) * self.plane_ids]

    def _start_exchange(self, k: int, buf, lo):
        """Halo of k planes for the pass with step k on the state in `buf`: my bottom / top k planes go down / up, theirs land
        in the planes just outside my slab.  Returns the requests (waited for right before the pass that needs them)."""
        d, P = self.dist, self.dist.P2POp
        ops = []
        if self.rank > 0:
            ops.append(P(d.isend, self._planes(buf, lo, self.z0, self.z0 + k), self.rank - 1))
            ops.append(P(d.irecv, self._planes(buf, lo, self.z0 - k, self.z0), self.rank - 1))
            self.bytes_received += k * self.plane_ids * 4
        if self.rank < self.world - 1:
            ops.append(P(d.isend, self._planes(buf, lo, self.z1 - k, self.z1), self.rank + 1))
            ops.append(P(d.irecv, self._planes(buf, lo, self.z1, self.z1 + k), self.rank + 1))
            self.bytes_received += k * self.plane_ids * 4
        return d.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def _wait(reqs):
        for r in reqs:
            r.wait()

    def jfa(self, algo=ALGO_TILED, fill=-math.inf, out=None):
        out = self.sdf if out is None else out
        G, n, z0, z1, pw = self.global_frame, self.global_frame.n, self.z0, self.z1, self.plane_words
        mask_start = (bool(self.wide) and len(self.wide) + len(self.narrow) > 1 and hasattr(self.be, "can_start_from_mask")
                      and self.be.can_start_from_mask(G, algo))
        lo, hi, (a, b) = self._window(mask_start)
        slab_words = self.words[z0 * pw:z1 * pw]
        npass = len(self.wide) + len(self.narrow)
        # ---- wide passes: ghost planes
        start = 0
        if mask_start:
            if self.border is None:
                self.border = self.be.empty_u32(G.words)
            self.be.surface(G, self.words, self.border)
            k, b0, b1 = self.wide[0]
            self.be.jfa_first_pass_window(G.slab(b0, b1), self.border, b, lo)
            a, b = b, a
            start = 1
        else:
            below = self.words[(lo - 1) * pw:lo * pw] if lo > 0 else None
            above = self.words[hi * pw:(hi + 1) * pw] if hi < n else None
            self.be.jfa_init(G.slab(lo, hi), self.words[lo * pw:hi * pw], below, above, a)
        for i in range(start, len(self.wide)):
            k, b0, b1 = self.wide[i]
            if i == npass - 1:                                        # one rank: the last pass is a wide one
                self.be.jfa_last_pass_window(G.slab(b0, b1), a, b, lo, slab_words, fill, out, algo)
                return out
            self.be.jfa_pass_window(G.slab(b0, b1), k, a, b, lo, algo)
            a, b = b, a
        # ---- narrow passes: halos from the adjacent ranks, the next pass's halo sent under this pass's interior planes
        pend = None
        for idx, k in enumerate(self.narrow):
            if pend is None:
                pend = self._start_exchange(k, a, lo)
            self._wait(pend)
            pend = None
            if idx == len(self.narrow) - 1:
                self.be.jfa_last_pass_window(self.frame, a, b, lo, slab_words, fill, out, algo)
                return out
            nk = self.narrow[idx + 1]
            nb = -(-nk // 8) * 8                                       # sub-slabs are cut at multiples of 8 planes
            if 2 * nb >= self.nz:
                self.be.jfa_pass_window(self.frame, k, a, b, lo, algo)
                pend = self._start_exchange(nk, b, lo)
            else:
                self.be.jfa_pass_window(G.slab(z0, z0 + nb), k, a, b, lo, algo)          # what the neighbours need next: first
                self.be.jfa_pass_window(G.slab(z1 - nb, z1), k, a, b, lo, algo)
                pend = self._start_exchange(nk, b, lo)
                self.be.jfa_pass_window(G.slab(z0 + nb, z1 - nb), k, a, b, lo, algo)     # the transfer runs under this
            a, b = b, a
        return out


# =============================================================================================
def make_pipeline(kind: str, engine, frame: Frame, rank: int, world: int, dist):
    """bench.py / callers: 'ghost' (no exchange), 'halo' (RCCL point-to-point halos before every pass) or 'hybrid' (ghost planes
    for the wide passes, overlapped halos for the narrow ones) on the HIP backend."""
    be = HipSlabBackend(engine)
    if kind == "ghost":
        return GhostSlabPipeline(be, frame, rank, world)
    if kind == "halo":
        return SlabPipeline(be, frame, rank, world, dist)
    if kind == "hybrid":
        return HybridSlabPipeline(be, frame, rank, world, dist)
    raise ValueError("unknown multi-GPU pipeline %r" % kind)
