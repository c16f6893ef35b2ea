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

Four pipelines feed a pass its planes z -+ k: SlabPipeline (halos before every pass), GhostSlabPipeline (ghost
planes recomputed, nothing exchanged), HybridSlabPipeline (ghost planes for the wide passes, halos for the narrow ones) and
TransposeSlabPipeline (planes dealt CYCLICALLY for every pass whose step is a multiple of the rank count -- no exchange, no ghost
planes -- then ONE all-to-all into slabs for the last log2(world) passes).

The compute backend is pluggable only so that the exchange logic can be exercised on CPU with gloo
in the test-suite; the product backend (HipSlabBackend) calls libvphip.so and nothing else.
"""
from __future__ import annotations

import math

import torch

from .capi import ALGO_TILED, Frame, Window


class IdWindow:
    """An id window (include/vphip.h, vp_jfa_window_*): `planes` id planes in the backend's own layout, held by one tensor."""

    def __init__(self, t, planes):
        self.t, self.planes = t, int(planes)


class HipSlabBackend:
    """Product backend: torch CUDA tensors for memory, libvphip.so (via Engine.ctx) for compute.  JFA state lives in id windows; the
    tile kernels that run on them start at n = 96 (smaller grids are not worth sharding: one GPU does 64^3 in 0.06 ms)."""

    def __init__(self, engine, poison=None):
        self.engine = engine
        self.ctx = engine.ctx
        self.device = engine.device
        self.poison = poison          # tests: a byte the word planes of a fresh window are filled with after the clear (see window())

    def check_frame(self, frame):
        if frame.n < 96:
            raise ValueError("n = %d: the Z-slab pipelines run the tile kernels (n >= 96); use one GPU below that" % frame.n)

    def empty_u32(self, n):
        return torch.empty(int(n), dtype=torch.int32, device=self.device)

    def empty_f32(self, n):
        return torch.empty(int(n), dtype=torch.float32, device=self.device)

    def voxelize(self, frame, words, d_xyz, d_tri, algo):
        self.ctx.voxelize(frame, words.data_ptr(), d_xyz.data_ptr(), d_xyz.shape[0], d_tri.data_ptr(), d_tri.shape[0], algo, False)

    def csg(self, a, b, op):
        self.ctx.csg(a.data_ptr(), b.data_ptr(), a.numel(), op)

    def surface(self, frame, words, border):
        self.ctx.surface(frame, words.data_ptr(), None, None, border.data_ptr())

    def can_start_from_mask(self, frame):
        return self.ctx.jfa_can_start_from_mask(frame, ALGO_TILED)

    @staticmethod
    def _p(t):
        return t.data_ptr() if t is not None else None

    # -- id windows ----------------------------------------------------------------------------
    def window(self, frame, planes):
        """A fresh window, every id "none".  The regions of the ghost / hybrid pipelines are rounded outwards to the 8-plane tile, and
        the excess planes of a pass read planes the pass before it never produced (ghost_regions): what they read is then "none" -- never
        memory nobody wrote.  `poison` (tests) overwrites the word planes with an arbitrary byte afterwards to show that the results do
        not depend on what those planes hold."""
        t = torch.empty(self.ctx.jfa_window_bytes(frame, planes), dtype=torch.uint8, device=self.device)
        self.ctx.jfa_window_clear(frame, Window.make(t.data_ptr(), t.numel(), planes, 0))
        if self.poison is not None:
            t[:planes * frame.n * frame.n * 4].fill_(int(self.poison) & 0xFF)
        return IdWindow(t, planes)

    def win_spans(self, frame, w, p0, p1):
        """the tensor slices that hold the planes [p0, p1) of a window: what a halo exchange sends / receives"""
        return [w.t[o:o + nb] for o, nb in self.ctx.jfa_window_span(frame, w.planes, p0, p1)]

    @staticmethod
    def _w(w, at):
        return Window.make(w.t.data_ptr(), w.t.numel(), w.planes, at)

    def win_init(self, region, words_region, below, above, w, at):
        self.ctx.jfa_window_init(region, words_region.data_ptr(), self._p(below), self._p(above), self._w(w, at))

    def win_first_pass(self, region, border_full, w, at):
        self.ctx.jfa_window_first_pass(region, border_full.data_ptr(), self._w(w, at))

    def win_first_two(self, frame, border_full, w):
        self.ctx.jfa_window_first_two(frame, border_full.data_ptr(), self._w(w, 0))

    def win_pass(self, region, k, w_in, w_out, at, stride=None):
        self.ctx.jfa_window_pass(region, k, self._w(w_in, at), self._w(w_out, at), stride)

    def win_last_pass(self, region, w_in, w_scratch, at, words_region, fill, sdf, stride=1):
        self.ctx.jfa_window_last_pass(region, self._w(w_in, at), self._w(w_scratch, at), words_region.data_ptr(), fill, sdf.data_ptr(), stride)


    # -- cyclic plane distribution (TransposeSlabPipeline) ----------------------------------------
    def cyclic_passes(self, frame, world):
        return self.ctx.jfa_cyclic_passes(frame, world)

    def win_first_two_cyclic(self, frame, border_full, w, world, rank):
        self.ctx.jfa_window_first_two_cyclic(frame, border_full.data_ptr(), self._w(w, 0), world, rank)

    def win_pass_cyclic(self, frame, k, w_in, w_out, world, rank):
        self.ctx.jfa_window_pass_cyclic(frame, k, self._w(w_in, 0), self._w(w_out, 0), world, rank)

    def win_interleave(self, frame, w_in, w_out, at, world, count):
        self.ctx.jfa_window_interleave(frame, self._w(w_in, 0), self._w(w_out, at), world, count)


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

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None):
        host_out = torch.empty(output.shape, dtype=output.dtype)
        self.dist.all_to_all_single(host_out, input.cpu(), output_split_sizes, input_split_sizes)
        output.copy_(host_out)

    def barrier(self):
        self.dist.barrier()


def hbm_bytes(obj) -> int:
    """bytes of device memory held by the tensors of a pipeline object (attributes, and lists / dicts of them), each storage once"""
    seen, total = set(), 0

    def walk(v):
        nonlocal total
        if isinstance(v, IdWindow):
            v = v.t
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
    """voxelize -> (CSG) -> JFA for the slab of this rank, halo planes exchanged point to point before every pass.

    A rank's two id windows hold [slab holding z - k | own slab | slab holding z + k] = 3 nz planes, the own slab in the middle.  The halos
    of the narrow passes (k <= nz/2) land in the k planes right below / above the slab (stride = k: consecutive planes).  For the wide
    passes (k >= nz) whole slabs of distant ranks land nz planes below / above the own planes and the tile kernel runs with stride = nz
    (include/vphip.h, vp_jfa_window_pass) -- one kernel, one buffer layout, and above n = 1024 five bytes per voxel on the wire."""

    def __init__(self, backend, frame: Frame, rank: int, world: int, dist):
        self.be = backend
        self.dist = dist
        self.rank, self.world = rank, world
        self.global_frame = frame
        self.be.check_frame(frame)
        self.z0, self.z1 = slab_range(frame.n, rank, world)
        self.frame = frame.slab(self.z0, self.z1)
        n = frame.n
        self.plane_words = n * n // 32            # bitmask words per plane
        self.nz = self.z1 - self.z0
        self.words = self.be.empty_u32(self.frame.words)
        self.planes = 3 * self.nz if world > 1 else self.nz
        self.at = self.nz if world > 1 else 0
        self.ids = [self.be.window(frame, self.planes) for _ in range(2)]
        self.sdf = self.be.empty_f32(self.frame.voxels)
        self.below = self.be.empty_u32(self.plane_words) if rank > 0 else None
        self.above = self.be.empty_u32(self.plane_words) if rank < world - 1 else None
        self.bytes_received = 0

    def describe(self):
        return "z-slab x%d, RCCL p2p halo exchange before every pass" % self.world

    def report(self):
        return {"pipeline": "halo", "slab_planes": self.nz, "window_planes": self.planes, "bytes_received_total": int(self.bytes_received),
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

    def _stride(self, k: int) -> int:
        return self.nz if (self.world > 1 and k >= self.nz) else k

    def _exchange_ids(self, k: int, src):
        d, P = self.dist, self.dist.P2POp
        G, stride = self.global_frame, self._stride(k)
        ops = []
        for s, t, side, g0, g1 in halo_plan(G.n, self.world, k):
            if s == t:
                continue
            if s == self.rank:
                for piece in self.be.win_spans(G, src, self.at + g0 - self.z0, self.at + g1 - self.z0):
                    ops.append(P(d.isend, piece, t))
            elif t == self.rank:
                # global plane g feeds my plane z = g + k (minus side) / g - k (plus side), which sits `stride` planes away from it
                dp = self.at + (g0 + k - self.z0) - stride if side == "minus" else self.at + (g0 - k - self.z0) + stride
                for piece in self.be.win_spans(G, src, dp, dp + (g1 - g0)):
                    ops.append(P(d.irecv, piece, s))
                    self.bytes_received += piece.numel() * piece.element_size()
        self._exchange(ops)

    def jfa(self, fill=-math.inf, out=None):
        out = self.sdf if out is None else out
        if self.world > 1:
            self._exchange_mask_planes()
        a, b = self.ids
        self.be.win_init(self.frame, self.words, self.below, self.above, a, self.at)
        k = self.global_frame.n // 2
        while k >= 1:                                             # jfa/sequential.cpp:72
            if self.world > 1:
                self._exchange_ids(k, a)
            if k == 1:                                            # last pass + id -> sdf conversion fused
                self.be.win_last_pass(self.frame, a, b, self.at, self.words, fill, out, self._stride(1))
                return out
            self.be.win_pass(self.frame, k, a, b, self.at, self._stride(k))
            a, b = b, a
            k //= 2
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


class GhostSlabPipeline:
    """Z-slab strong scaling WITHOUT halo exchange.

    Measured on MI355X one JFA pass costs ~1.6 us per 512^2 plane, while moving that plane (1 MiB) to a
    peer over one xGMI link costs ~20 us: recomputing ghost planes is an order of magnitude cheaper than
    exchanging them.  Every rank therefore voxelizes the whole grid (0.06 ms at n = 512), runs the first two
    passes as the one whole-grid launch of the single-GPU path (its second pass would cover 35 % of the grid or
    more on any rank of 2 .. 8 slabs: the break-even of the measured kernel times, profiles/r03/slab_scaling_*.txt)
    and every later pass i on its slab widened by the reach of the passes after it (ghost_regions).  The regions
    shrink to the bare slab at k = 1; the result is the slab's part of the single-GPU result, bit for
    bit, with zero bytes exchanged.  The two id windows hold the whole grid (a plane sits at its global index):
    2 x 4 n^3 B up to n = 1024, 2 x 5 n^3 B above (2 x 40 GiB at n = 2048).
    """

    def __init__(self, backend, frame: Frame, rank: int, world: int):
        self.be = backend
        self.rank, self.world = rank, world
        self.global_frame = frame
        self.be.check_frame(frame)
        self.z0, self.z1 = slab_range(frame.n, rank, world)
        self.frame = frame.slab(self.z0, self.z1)
        self.regions = ghost_regions(frame.n, rank, world)
        self.words = self.be.empty_u32(frame.words)                 # whole grid
        self.ids = [self.be.window(frame, frame.n) for _ in range(2)]
        self.sdf = self.be.empty_f32(self.frame.voxels)             # own slab only
        self.planes_computed = sum(b1 - b0 for _, b0, b1 in self.regions)
        self.border = self.be.empty_u32(frame.words)

    def describe(self):
        return "z-slab x%d, ghost planes recomputed, no data-path exchange" % self.world

    def report(self):
        n, passes = self.global_frame.n, len(self.regions)
        return {"pipeline": "ghost", "slab_planes": self.z1 - self.z0, "regions": [[k, b0, b1] for k, b0, b1 in self.regions],
                "plane_passes_this_rank": int(self.planes_computed), "plane_passes_one_gpu": n * passes,
                "plane_pass_ratio": round(n * passes / self.planes_computed, 3), "bytes_exchanged": 0, "bytes_received_total": 0,
                "id_window_bytes": int(self.ids[0].t.numel() * self.ids[0].t.element_size()),
                "hbm_bytes_this_rank": hbm_bytes(self)}

    def voxelize(self, d_xyz, d_tri, algo=ALGO_TILED, out=None):
        out = self.words if out is None else out
        self.be.voxelize(self.global_frame, out, d_xyz, d_tri, algo)
        return out

    def csg(self, other, op: int):
        self.be.csg(self.words, other, op)
        return self.words

    def jfa(self, fill=-math.inf, out=None):
        """border mask, passes n/2 + n/4 over the whole grid in one launch, every later pass on its region"""
        out = self.sdf if out is None else out
        a, b = self.ids
        G = self.global_frame
        self.be.surface(G, self.words, self.border)
        self.be.win_first_two(G, self.border, a)
        pw = G.n * G.n // 32
        last = len(self.regions) - 1
        for i, (k, b0, b1) in enumerate(self.regions):
            if i < 2:
                continue
            region = G.slab(b0, b1)
            if i == last:
                self.be.win_last_pass(region, a, b, b0, self.words[b0 * pw:b1 * pw], fill, out)
                return out
            self.be.win_pass(region, k, a, b, b0)
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
        self.be.check_frame(frame)
        self.z0, self.z1 = slab_range(frame.n, rank, world)
        self.nz = self.z1 - self.z0
        self.frame = frame.slab(self.z0, self.z1)
        self.wide, self.narrow = hybrid_plan(frame.n, rank, world)
        self.plane_words = frame.n * frame.n // 32
        self.words = self.be.empty_u32(frame.words)                  # whole grid: every rank rasterises it (0.06 ms at n = 512)
        self.sdf = self.be.empty_f32(self.frame.voxels)
        self.border = None
        self.mask_start = bool(self.wide) and len(self.wide) + len(self.narrow) > 1 and self.be.can_start_from_mask(frame)
        self.window = list(self._window_planes(self.mask_start))     # global planes [lo, hi) the two id windows hold
        self.ids = [self.be.window(frame, self.window[1] - self.window[0]) for _ in range(2)]
        self.bytes_received = 0
        self.planes_computed = sum(b1 - b0 for _, b0, b1 in self.wide) + self.nz * len(self.narrow)

    def describe(self):
        return "z-slab x%d, hybrid: ghost planes for k > nz/2, p2p halos under the interior planes for k <= nz/2" % self.world

    def report(self):
        n, passes = self.global_frame.n, len(self.wide) + len(self.narrow)
        return {"pipeline": "hybrid", "slab_planes": self.nz, "wide_regions": [[k, b0, b1] for k, b0, b1 in self.wide],
                "narrow_steps": list(self.narrow), "plane_passes_this_rank": int(self.planes_computed),
                "plane_passes_one_gpu": n * passes, "plane_pass_ratio": round(n * passes / self.planes_computed, 3),
                "bytes_received_total": int(self.bytes_received), "id_buffer_planes": self.window,
                "hbm_bytes_this_rank": hbm_bytes(self)}

    def voxelize(self, d_xyz, d_tri, algo=ALGO_TILED, out=None):
        out = self.words if out is None else out
        self.be.voxelize(self.global_frame, out, d_xyz, d_tri, algo)
        return out

    def csg(self, other, op: int):
        self.be.csg(self.words, other, op)
        return self.words

    # -- the window: planes [lo, hi) of the id volume ------------------------------------------
    def _window_planes(self, mask_start: bool):
        n, H = self.global_frame.n, (self.nz // 2 if self.world > 1 else 0)
        lo, hi = max(0, self.z0 - H), min(n, self.z1 + H)            # room for the halos of the narrow passes
        for i, (k, b0, b1) in enumerate(self.wide):
            lo, hi = min(lo, b0), max(hi, b1)
            if i > 0 or not mask_start:                               # every pass that reads ids reads k planes beyond its region
                lo, hi = min(lo, max(0, b0 - k)), max(hi, min(n, b1 + k))
        return lo, hi

    def _start_exchange(self, k: int, w):
        """Halo of k planes for the pass with step k on the state in `w`: my bottom / top k planes go down / up, theirs land
        in the planes just outside my slab.  Returns the requests (waited for right before the pass that needs them)."""
        d, P = self.dist, self.dist.P2POp
        G, lo = self.global_frame, self.window[0]
        ops = []

        def planes(g0, g1):
            return self.be.win_spans(G, w, g0 - lo, g1 - lo)

        if self.rank > 0:
            ops += [P(d.isend, t, self.rank - 1) for t in planes(self.z0, self.z0 + k)]
            for t in planes(self.z0 - k, self.z0):
                ops.append(P(d.irecv, t, self.rank - 1))
                self.bytes_received += t.numel() * t.element_size()
        if self.rank < self.world - 1:
            ops += [P(d.isend, t, self.rank + 1) for t in planes(self.z1 - k, self.z1)]
            for t in planes(self.z1, self.z1 + k):
                ops.append(P(d.irecv, t, self.rank + 1))
                self.bytes_received += t.numel() * t.element_size()
        return d.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def _wait(reqs):
        for r in reqs:
            r.wait()

    def jfa(self, fill=-math.inf, out=None):
        out = self.sdf if out is None else out
        G, n, z0, z1, pw = self.global_frame, self.global_frame.n, self.z0, self.z1, self.plane_words
        lo, hi = self.window
        a, b = self.ids
        slab_words = self.words[z0 * pw:z1 * pw]
        npass = len(self.wide) + len(self.narrow)
        # ---- wide passes: ghost planes
        start = 0
        if self.mask_start:
            if self.border is None:
                self.border = self.be.empty_u32(G.words)
            self.be.surface(G, self.words, self.border)
            k, b0, b1 = self.wide[0]
            self.be.win_first_pass(G.slab(b0, b1), self.border, b, b0 - lo)
            a, b = b, a
            start = 1
        else:
            below = self.words[(lo - 1) * pw:lo * pw] if lo > 0 else None
            above = self.words[hi * pw:(hi + 1) * pw] if hi < n else None
            self.be.win_init(G.slab(lo, hi), self.words[lo * pw:hi * pw], below, above, a, 0)
        for i in range(start, len(self.wide)):
            k, b0, b1 = self.wide[i]
            if i == npass - 1:                                        # one rank: the last pass is a wide one
                self.be.win_last_pass(G.slab(b0, b1), a, b, b0 - lo, slab_words, fill, out)
                return out
            self.be.win_pass(G.slab(b0, b1), k, a, b, b0 - lo)
            a, b = b, a
        # ---- narrow passes: halos from the adjacent ranks, the next pass's halo sent under this pass's interior planes
        pend = None
        at = z0 - lo
        for idx, k in enumerate(self.narrow):
            if pend is None:
                pend = self._start_exchange(k, a)
            self._wait(pend)
            pend = None
            if idx == len(self.narrow) - 1:
                self.be.win_last_pass(self.frame, a, b, at, slab_words, fill, out)
                return out
            nk = self.narrow[idx + 1]
            nb = -(-nk // 8) * 8                                       # sub-slabs are cut at multiples of 8 planes
            if 2 * nb >= self.nz:
                self.be.win_pass(self.frame, k, a, b, at)
                pend = self._start_exchange(nk, b)
            else:
                self.be.win_pass(G.slab(z0, z0 + nb), k, a, b, at)                       # what the neighbours need next: first
                self.be.win_pass(G.slab(z1 - nb, z1), k, a, b, at + (z1 - nb - z0))
                pend = self._start_exchange(nk, b)
                self.be.win_pass(G.slab(z0 + nb, z1 - nb), k, a, b, at + nb)             # the transfer runs under this
            a, b = b, a
        return out


# =============================================================================================
# Transposed: cyclic planes while the steps are multiples of the rank count, one all-to-all, slabs for the rest
# =============================================================================================
def cyclic_passes(n: int, world: int, min_n: int = 96) -> int:
    """Passes of the sequence n/2, n/4, ... (jfa/sequential.cpp:72) whose step is a multiple of `world`, counted from the first: with the
    planes dealt cyclically (plane z on rank z mod world) such a pass finds the planes z - k, z, z + k (:92-94) of every plane a rank owns
    on that rank.  0 where the distribution does not apply: fewer than the two passes of the fused start, a rank count that is not a power
    of two, slabs that are not a multiple of 8 planes, grids below the tile kernels' range (min_n; the numpy backend of the tests has
    none).  (= vp_jfa_cyclic_passes)"""
    if n < min_n or world < 2 or world & (world - 1) or n % world or (n // world) % 8:
        return 0
    c, k = 0, n // 2
    while k >= 1 and k % world == 0:
        c, k = c + 1, k // 2
    return c if c >= 2 else 0


def transpose_plan(n: int, rank: int, world: int, min_n: int = 96):
    """What rank `rank` does in the transposed pipeline, or None where it does not apply (cyclic_passes == 0):
      cyclic   the steps that run on the cyclic distribution (local plane l = global plane rank + l * world), no exchange
      regions  [(k, b0, b1)] the remaining steps on the slab widened by the reach of the later ones -- ghost_regions restricted to them
      recv     [t0, t1): the global planes the all-to-all delivers to this rank -- the slab widened by the reach g of ALL remaining steps,
               rounded outwards to a multiple of world so that every source holds the same local range [t0 / world, t1 / world) of it
      window   [lo, hi): the planes the two slab-phase id windows hold: recv and what the (outward-rounded) regions read beyond it"""
    c = cyclic_passes(n, world, min_n)
    if c == 0:
        return None
    regs = ghost_regions(n, rank, world)
    ks = [k for k, _, _ in regs]
    z0, z1 = slab_range(n, rank, world)
    g = sum(ks[c:])
    t0, t1 = max(0, (z0 - g) // world * world), min(n, -(-(z1 + g) // world) * world)
    lo = min([t0] + [max(0, b0 - k) for k, b0, _ in regs[c:]])
    hi = max([t1] + [min(n, b1 + k) for k, _, b1 in regs[c:]])
    return {"cyclic": ks[:c], "regions": regs[c:], "recv": (t0, t1), "window": (lo, hi)}


class TransposeSlabPipeline:
    """Z-slab strong scaling with ONE exchange.

    The reference's pass with step k reads the planes z - k, z, z + k of a voxel's plane z and nothing else (jfa/sequential.cpp:72,
    :92-94).  Phase A: the planes are dealt cyclically -- rank r keeps the planes z = r (mod world) in two windows of n / world planes --
    and every pass whose step is a multiple of world (all of them down to k = world on a power-of-two grid) runs without any exchange or
    ghost plane: each rank does exactly 1 / world of the pass, on chains of full length.  Every rank voxelizes the whole grid and takes
    its border mask, as the ghost pipeline does (0.1 ms at n = 1024); the fused start writes only the rank's planes.  Phase B: one
    all-to-all -- the planes of rank t's widened slab that rank s holds are a CONTIGUOUS range of s's local planes -- re-deals the state
    into slabs widened by the reach of the remaining steps (world - 1 planes: 7 for 8 ranks, rounded to 8), a local kernel weaves the
    received chunks into consecutive planes, and the passes k < world run on the widened slab exactly like the last regions of the ghost
    pipeline.  Per rank at n = 1024 x 8: 1,280 + 2 x 16 extra plane-passes against 3,216 (ghost), 0.49 GiB received against 3.5 GiB (halo),
    id state 2 x 0.5 (cyclic) + 2 x 0.56 (send, staging: not with exchange = "p2p") + 2 x 0.6 GiB (slab phase).  Grids whose step sequence leaves the multiples of world early (sides that are not powers of two) simply
    switch to the slab phase earlier, with a wider margin; where fewer than two passes qualify the pipeline IS the ghost pipeline."""

    def __init__(self, backend, frame: Frame, rank: int, world: int, dist, exchange: str = "a2a"):
        """exchange: "a2a" = one all_to_all_single of packed chunks + the weave (the default); "p2p" = every plane sent point to point
        straight from the window of the cyclic phase to its final place in the receiver's slab window (one batch of isend / irecv: no send
        buffer, no staging buffer, no weave -- at the price of (world - 1) x count messages per rank instead of one collective)"""
        if exchange not in ("a2a", "p2p"):
            raise ValueError("exchange must be 'a2a' or 'p2p'")
        self.be, self.dist, self.exchange_kind = backend, dist, exchange
        self.rank, self.world = rank, world
        self.global_frame = frame
        self.be.check_frame(frame)
        self.z0, self.z1 = slab_range(frame.n, rank, world)
        self.frame = frame.slab(self.z0, self.z1)
        min_n = getattr(backend, "tile_min_n", 96)
        self.plan = transpose_plan(frame.n, rank, world, min_n)
        self.fallback = None
        self.bytes_received = 0
        if self.plan is None:                                      # nothing to deal cyclically: ghost planes
            self.fallback = GhostSlabPipeline(backend, frame, rank, world)
            self.words, self.sdf = self.fallback.words, self.fallback.sdf
            return
        assert self.be.cyclic_passes(frame, world) == len(self.plan["cyclic"]), "the library counts the cyclic passes differently"
        n = frame.n
        self.nzl = n // world                                       # planes of the cyclic share
        self.words = self.be.empty_u32(frame.words)                 # whole grid: every rank rasterises it
        self.border = self.be.empty_u32(frame.words)
        self.sdf = self.be.empty_f32(self.frame.voxels)
        t0, t1 = self.plan["recv"]
        lo, hi = self.plan["window"]
        self.count = (t1 - t0) // world                             # planes every source sends me
        # what I send to rank t: my local planes [a, b) = the planes = rank (mod world) of t's recv range
        self.send_ranges = []
        for t in range(world):
            p = transpose_plan(n, t, world, min_n)
            self.send_ranges.append((p["recv"][0] // world, p["recv"][1] // world))
        # id windows, allocated at first use and kept (a steady-state step allocates nothing): the two of the cyclic phase, the packed
        # send buffer, the staging buffer the all-to-all fills, the two of the slab phase
        self._planes = {"cyc0": self.nzl, "cyc1": self.nzl, "send": sum(b - a for a, b in self.send_ranges), "staging": self.count * world,
                        "ids0": hi - lo, "ids1": hi - lo}
        self.win = {}
        self.planes_computed = self.nzl * len(self.plan["cyclic"]) + sum(b1 - b0 for _, b0, b1 in self.plan["regions"])

    def _w(self, name):
        if name not in self.win:
            self.win[name] = self.be.window(self.global_frame, self._planes[name])
        return self.win[name]

    def release(self, *names):
        """drop id windows (tests that walk the ranks of a large job one after the other on one GPU)"""
        for nm in names or list(self.win):
            self.win.pop(nm, None)

    def describe(self):
        if self.fallback is not None:
            return self.fallback.describe() + " (no step of this grid is a multiple of the rank count twice: transposed = ghost)"
        how = "one RCCL all-to-all" if self.exchange_kind == "a2a" else "one batch of RCCL point-to-point planes, placed directly"
        return "x%d transposed: planes dealt cyclically for the %d passes k >= %d (no exchange), %s, z-slabs for k < %d" % (
            self.world, len(self.plan["cyclic"]), self.plan["cyclic"][-1], how, self.plan["cyclic"][-1])

    def report(self):
        if self.fallback is not None:
            return dict(self.fallback.report(), pipeline="transpose->ghost")
        n, passes = self.global_frame.n, len(self.plan["cyclic"]) + len(self.plan["regions"])
        return {"pipeline": "transpose", "exchange": self.exchange_kind, "slab_planes": self.z1 - self.z0, "cyclic_steps": list(self.plan["cyclic"]),
                "slab_regions": [[k, b0, b1] for k, b0, b1 in self.plan["regions"]], "recv_planes": list(self.plan["recv"]),
                "window_planes": list(self.plan["window"]), "plane_passes_this_rank": int(self.planes_computed),
                "plane_passes_one_gpu": n * passes, "plane_pass_ratio": round(n * passes / self.planes_computed, 3),
                "bytes_received_total": int(self.bytes_received), "hbm_bytes_this_rank": hbm_bytes(self)}

    def voxelize(self, d_xyz, d_tri, algo=ALGO_TILED, out=None):
        if self.fallback is not None:
            return self.fallback.voxelize(d_xyz, d_tri, algo, out)
        out = self.words if out is None else out
        self.be.voxelize(self.global_frame, out, d_xyz, d_tri, algo)
        return out

    def csg(self, other, op: int):
        self.be.csg(self.words, other, op)
        return self.words

    # -- the four stages of a JFA (jfa() below runs them in order; tests drive them one by one) -----------------------------
    def phase_a(self):
        """cyclic planes, no exchange: border mask of the whole grid, the fused start for the rank's planes, every further step that is a
        multiple of the rank count; returns the window that holds the result"""
        G, be, w, r = self.global_frame, self.be, self.world, self.rank
        a, b = self._w("cyc0"), self._w("cyc1")
        be.surface(G, self.words, self.border)
        be.win_first_two_cyclic(G, self.border, a, w, r)
        for k in self.plan["cyclic"][2:]:
            be.win_pass_cyclic(G, k, a, b, w, r)
            a, b = b, a
        return a

    def pack(self, src):
        """my planes of every rank's widened slab, one contiguous piece per destination (the ranges overlap -- the margins of neighbouring
        slabs -- and an all-to-all wants disjoint pieces)"""
        G, be = self.global_frame, self.be
        send, at = self._w("send"), 0
        for a, b in self.send_ranges:
            for dst, piece in zip(be.win_spans(G, send, at, at + (b - a)), be.win_spans(G, src, a, b)):
                dst.copy_(piece)
            at += b - a

    def exchange(self):
        """THE exchange of the job: one all_to_all_single (two above n = 1024: the word part and the byte part of the windows).  Chunk s of
        the staging window = the planes t0 + s, t0 + s + world, ... of my widened slab, as rank s kept them"""
        G, be, w = self.global_frame, self.be, self.world
        send, staging = self._w("send"), self._w("staging")
        outs = be.win_spans(G, staging, 0, staging.planes)
        ins = be.win_spans(G, send, 0, send.planes)
        for o, i in zip(outs, ins):
            per_plane = o.numel() // staging.planes
            self.dist.all_to_all_single(o, i, [self.count * per_plane] * w, [(b - a) * per_plane for a, b in self.send_ranges])
            self.bytes_received += (w - 1) * self.count * per_plane * o.element_size()

    def exchange_p2p(self, src):
        """the re-deal without a send buffer, a staging buffer or a weave: plane t0 + s + j world of my widened slab is received from rank s
        straight into its place in the slab window, my planes of every other rank's slab are sent straight from the window of the cyclic
        phase -- one batch of point-to-point operations (posted in the same plane order on both sides); my own planes: one strided copy"""
        G, be, w, r = self.global_frame, self.be, self.world, self.rank
        d, P = self.dist, self.dist.P2POp
        lo, _ = self.plan["window"]
        dst = self._w("ids0")
        at = self.plan["recv"][0] - lo
        ops = []
        for t in range(w):                                           # sends: to rank t its planes [a, b) of my window, in order
            if t == r:
                continue
            a, b = self.send_ranges[t]
            for l in range(a, b):
                ops += [P(d.isend, piece, t) for piece in be.win_spans(G, src, l, l + 1)]
        for s_ in range(w):                                          # receives: chunk s = the planes at + s, at + s + world, ...
            if s_ == r:
                continue
            for j in range(self.count):
                for piece in be.win_spans(G, dst, at + s_ + j * w, at + s_ + j * w + 1):
                    ops.append(P(d.irecv, piece, s_))
                    self.bytes_received += piece.numel() * piece.element_size()
        reqs = d.batch_isend_irecv(ops) if ops else []
        a, b = self.send_ranges[r]                                   # my own planes of my own slab
        for dpart, spart in zip(be.win_spans(G, dst, 0, dst.planes), be.win_spans(G, src, 0, src.planes)):
            dpart.view(dst.planes, -1)[at + r:at + r + w * self.count:w].copy_(spart.view(src.planes, -1)[a:b])
        for q in reqs:
            q.wait()

    def phase_b(self, fill=-math.inf, out=None, weave=True):
        """the weave into consecutive planes (exchange "a2a"), then the remaining steps on the widened slab (as the last regions of the ghost
        pipeline)"""
        out = self.sdf if out is None else out
        G, be = self.global_frame, self.be
        pw = G.n * G.n // 32
        lo, _ = self.plan["window"]
        a, b = self._w("ids0"), self._w("ids1")
        if weave:
            be.win_interleave(G, self._w("staging"), a, self.plan["recv"][0] - lo, self.world, self.count)
        regs = self.plan["regions"]
        for i, (k, b0, b1) in enumerate(regs):
            region = G.slab(b0, b1)
            if i == len(regs) - 1:
                be.win_last_pass(region, a, b, b0 - lo, self.words[b0 * pw:b1 * pw], fill, out)
                return out
            be.win_pass(region, k, a, b, b0 - lo)
            a, b = b, a
        return out

    def jfa(self, fill=-math.inf, out=None):
        if self.fallback is not None:
            return self.fallback.jfa(fill, out)
        if self.exchange_kind == "p2p":
            self.exchange_p2p(self.phase_a())
            return self.phase_b(fill, out, weave=False)
        self.pack(self.phase_a())
        self.exchange()
        return self.phase_b(fill, out)


# =============================================================================================
def make_pipeline(kind: str, engine, frame: Frame, rank: int, world: int, dist):
    """bench.py / callers: 'ghost' (no exchange), 'halo' (RCCL point-to-point halos before every pass), 'hybrid' (ghost planes
    for the wide passes, overlapped halos for the narrow ones) or 'transpose' (cyclic planes, one RCCL all-to-all, slabs) on the HIP backend."""
    be = HipSlabBackend(engine)
    if kind == "ghost":
        return GhostSlabPipeline(be, frame, rank, world)
    if kind == "halo":
        return SlabPipeline(be, frame, rank, world, dist)
    if kind == "hybrid":
        return HybridSlabPipeline(be, frame, rank, world, dist)
    if kind == "transpose":
        return TransposeSlabPipeline(be, frame, rank, world, dist)
    if kind == "transpose-p2p":
        return TransposeSlabPipeline(be, frame, rank, world, dist, exchange="p2p")
    raise ValueError("unknown multi-GPU pipeline %r" % kind)
