"""Dev tool (one GPU): what the Z-slab pipelines cost per rank at the sizes the north star shards.

  python tools/slab_scaling.py 512 | 1024 | 2048

ghost planes (no exchange): every rank of a G-GPU job is run on THIS GPU with the real kernels and timed; ranks share
  nothing, so the job time on G GPUs is the slowest rank -- a measurement, not a model.  Also printed: plane-passes per
  rank against the single-GPU count (the plane-pass ratio: every plane-pass counted alike -- NOT a ceiling on the speedup, since the
  fused whole-grid launch of the first two passes is cheaper per plane than the passes it replaces).
hybrid (ghost planes for k > nz/2, halos of the adjacent ranks for k <= nz/2): plane-passes per rank and bytes received per side are
  exact (slab.hybrid_plan); times are a MODEL: plane-passes x the measured single-GPU time per plane-pass, + the halo bytes of one
  side at an assumed per-link rate, once fully hidden under the interior planes and once not hidden at all.
transposed (cyclic planes for the steps that are multiples of G, one all-to-all, slabs for the rest): per-rank COMPUTE is measured on this
  GPU with the real kernels on real data -- the cyclic phase + the pack of the send buffer, then the weave + the slab phase on a staging
  buffer filled with exactly the planes the all-to-all would deliver (cut from the whole-grid state after the cyclic steps, computed here);
  bytes received per rank are exact; the exchange time is a MODEL (bytes / an assumed aggregate ingest rate).
halo exchange: bytes each rank must RECEIVE over xGMI per job (exact, from slab.halo_plan) and the time that takes at two
  assumed per-GPU ingest rates -- a model (no multi-GPU box is reachable from here); compute per rank is the measured
  single-GPU time / G at best.
"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
from cuda_mesh_voxelization_amd.capi import Window
from cuda_mesh_voxelization_amd.slab import (GhostSlabPipeline, HipSlabBackend, HybridSlabPipeline, SlabPipeline, TransposeSlabPipeline, halo_plan, ghost_regions,
                                             hybrid_plan, transpose_plan)
only = sys.argv[2].split(",") if len(sys.argv) > 2 else ["ghost", "transpose", "halo", "hybrid"]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
refine = 24 if n <= 1024 else 192
xyz, tri = M.bunny(refine); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
S = eng.ctx.jfa_state_bytes(fr, ALGO_TILED); passes = int(math.log2(n))
reps = 10 if n <= 512 else 3

def timeit(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

g = eng.new_grid(fr); sdf = torch.empty(fr.voxels, dtype=torch.float32, device=eng.device)
def single():
    eng.voxelize(fr, dx, dt, out=g); eng.jfa(fr, g, out=sdf)
t1 = timeit(single, reps)
del g, sdf; eng._work = None; torch.cuda.empty_cache()
print("n = %d, %d faces, id bytes S = %d, %d passes; 1 GPU: %.3f ms per job" % (n, tri.shape[0], S, passes, t1))
print("\nghost planes (measured per rank on one GPU; job = slowest rank; nothing is exchanged)")
print("  G   job ms  speedup  efficiency  plane-passes/rank (1 GPU: %d)  pp-ratio  per-rank ms" % (n * passes))
for world in ((2, 4, 8) if "ghost" in only else ()):
    ts, pp = [], []
    for r in range(world):
        pipe = GhostSlabPipeline(HipSlabBackend(eng), fr, r, world)
        def step(): pipe.voxelize(dx, dt); pipe.jfa()
        ts.append(timeit(step, max(2, reps // 2))); pp.append(pipe.planes_computed)
        del pipe; torch.cuda.empty_cache()
    print("  %d  %7.3f  %6.2fx  %9.0f%%  %8d                        %5.2fx   %s"
          % (world, max(ts), t1 / max(ts), 100 * t1 / max(ts) / world, max(pp), n * passes / max(pp), " ".join("%.2f" % t for t in ts)))

# ---------------------------------------------------------------------------------------------- transposed
class NoDist:
    def all_to_all_single(self, *a, **k): pass

if "transpose" in only:
    print("\ntransposed (per-rank compute measured on this GPU, real kernels, real data; exchange: exact bytes, time MODELLED)")
    print("  G   cyclic steps   slab steps   compute ms: slowest rank (phase A + pack | weave + phase B)   <= 1.25 t1/G ?   GiB received/rank   "
          "+ exchange @ 300 / 150 GB/s into the rank   speedup vs 1 GPU   per-rank ms")
    ctx = eng.ctx
    words = eng.voxelize(fr, dx, dt)
    border = torch.empty_like(words)
    ctx.surface(fr, words.data_ptr(), None, None, border.data_ptr())
    for world in (2, 4, 8):
        plan0 = transpose_plan(n, 0, world)
        if plan0 is None:
            print("  %d   (no cyclic distribution for this grid)" % world); continue
        c = len(plan0["cyclic"])
        # the whole-grid state after the cyclic steps: what the all-to-all cuts the staging buffers from
        whole = [torch.empty(ctx.jfa_window_bytes(fr, n), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        W = lambda t: Window.make(t.data_ptr(), t.numel(), n, 0)
        ctx.jfa_window_first_two(fr, border.data_ptr(), W(whole[0]))
        cur, k = 0, n // 8
        for _ in range(c - 2):
            ctx.jfa_window_pass(fr, k, W(whole[cur]), W(whole[cur ^ 1])); cur ^= 1; k //= 2
        state = whole[cur]; del whole
        vox = n * n * n
        sw = state[:vox * 4].view(n, n * n * 4); sb = state[vox * 4:vox * 5].view(n, n * n) if n > 1024 else None
        ta, tb, rx, ta2, tb2 = [], [], [], [], []
        for r in range(world):
            pipe = TransposeSlabPipeline(HipSlabBackend(eng), fr, r, world, NoDist())
            def stepA(): pipe.voxelize(dx, dt); pipe.pack(pipe.phase_a())
            ta.append(timeit(stepA, max(2, reps // 2)))
            t0, t1_ = pipe.plan["recv"]
            staging = pipe._w("staging")
            parts = pipe.be.win_spans(fr, staging, 0, staging.planes)
            for part, src in zip(parts, (sw, sb)):
                part.view(world, pipe.count, -1).copy_(src[t0:t1_].view(pipe.count, world, -1).transpose(0, 1))
            def stepB(): pipe.phase_b()
            tb.append(timeit(stepB, max(2, reps // 2)))
            pipe.exchange(); rx.append(pipe.bytes_received)      # (NoDist: only the byte count)
            # the point-to-point form of the exchange: no pack, no weave -- the slab window filled with the same planes, directly
            def stepA2(): pipe.voxelize(dx, dt); pipe.phase_a()
            ta2.append(timeit(stepA2, max(2, reps // 2)))
            ids0 = pipe._w("ids0"); at = t0 - pipe.plan["window"][0]
            for part, src in zip(pipe.be.win_spans(fr, ids0, 0, ids0.planes), (sw, sb)):
                part.view(ids0.planes, -1)[at:at + (t1_ - t0)].copy_(src[t0:t1_])
            def stepB2(): pipe.phase_b(weave=False)
            tb2.append(timeit(stepB2, max(2, reps // 2)))
            del pipe, staging, parts, ids0; torch.cuda.empty_cache()
        tot = [a + b for a, b in zip(ta, tb)]
        i = max(range(world), key=lambda j: tot[j])
        b = max(rx)
        x300, x150 = b / 300e9 * 1e3, b / 150e9 * 1e3
        print("  %d   %-14s %-12s %8.3f (%.3f | %.3f)   %s (%.3f)   %6.3f   %8.3f / %8.3f ms   %.2fx / %.2fx (compute only: %.2fx)   %s"
              % (world, "%d..%d" % (plan0["cyclic"][0], plan0["cyclic"][-1]), ",".join(str(k_) for k_, _, _ in plan0["regions"]), tot[i], ta[i], tb[i],
                 "yes" if tot[i] <= 1.25 * t1 / world else "NO", 1.25 * t1 / world, b / 2**30, tot[i] + x300, tot[i] + x150,
                 t1 / (tot[i] + x300), t1 / (tot[i] + x150), t1 / tot[i], " ".join("%.2f" % t for t in tot)))
        tot2 = [a + b for a, b in zip(ta2, tb2)]
        i2 = max(range(world), key=lambda j: tot2[j])
        print("      exchange \"p2p\" (planes placed directly: no pack, no weave): compute %8.3f (%.3f | %.3f)   %s   compute only: %.2fx   %s"
              % (tot2[i2], ta2[i2], tb2[i2], "yes" if tot2[i2] <= 1.25 * t1 / world else "NO", t1 / tot2[i2], " ".join("%.2f" % t for t in tot2)))
        del state, sw, sb; torch.cuda.empty_cache()
    del words, border; torch.cuda.empty_cache()

if "halo" not in only and "hybrid" not in only:
    sys.exit(0)
print("\nhalo exchange (bytes received per rank and job: exact; times: MODEL at an assumed per-GPU ingest rate)")
print("  G   max GiB received/rank   @150 GB/s   @400 GB/s   + compute >= t1/G   vs 1 GPU")
plane = n * n * S
for world in (2, 4, 8):
    rx = [0] * world
    k = n // 2
    while k >= 1:
        for s_, t_, side, g0, g1 in halo_plan(n, world, k):
            if s_ != t_: rx[t_] += (g1 - g0) * plane
        k //= 2
    b = max(rx)
    t150, t400 = b / 150e9 * 1e3, b / 400e9 * 1e3
    print("  %d   %8.2f               %8.2f ms %8.2f ms   %8.2f ms         %.2fx .. %.2fx"
          % (world, b / 2**30, t150, t400, t1 / world, t1 / (t150 + t1 / world), t1 / (t400 + t1 / world)))

class NullDist:
    """No transfers at all: the halo planes keep whatever they held, so the RESULT of such a run is wrong -- its kernel sequence and
    per-rank compute time are those of the real job (the transfers are what the model below adds)."""
    isend, irecv = "isend", "irecv"
    class P2POp:
        def __init__(self, op, tensor, peer): pass
    def batch_isend_irecv(self, ops): return []

print("\nhalo pipeline, compute only (every rank measured on this GPU with the real kernels on its [slab - k | slab | slab + k] windows, transfers left out)")
print("  G   slowest rank ms   speedup if transfers were free   per-rank ms")
for world in (2, 4, 8):
    ts = []
    for r in range(world):
        pipe = SlabPipeline(HipSlabBackend(eng), fr, r, world, NullDist())
        def step(): pipe.voxelize(dx, dt); pipe.jfa()
        ts.append(timeit(step, max(2, reps // 2)))
        del pipe; torch.cuda.empty_cache()
    print("  %d   %8.3f          %5.2fx                            %s" % (world, max(ts), t1 / max(ts), " ".join("%.2f" % t for t in ts)))

print("\nhybrid, compute only (every rank measured on this GPU with the real kernels and sub-slab launches, transfers left out)")
print("  G   slowest rank ms   speedup if transfers were free   per-rank ms")
hyb = {}
for world in (2, 4, 8):
    ts = []
    for r in range(world):
        pipe = HybridSlabPipeline(HipSlabBackend(eng), fr, r, world, NullDist())
        def step(): pipe.voxelize(dx, dt); pipe.jfa()
        ts.append(timeit(step, max(2, reps // 2)))
        del pipe; torch.cuda.empty_cache()
    hyb[world] = max(ts)
    print("  %d   %8.3f          %5.2fx                            %s" % (world, max(ts), t1 / max(ts), " ".join("%.2f" % t for t in ts)))

print("\nhybrid (plane-passes and bytes: exact; times: MODEL -- per plane-pass = 1-GPU job / (n x passes) = %.2f us)" % (t1 * 1e3 / (n * passes)))
print("  G   plane-passes/rank  pp-ratio  id-buffer planes   GiB received per side   compute ms (measured)   + halos @150 GB/s per link: hidden .. exposed   vs 1 GPU")
for world in (2, 4, 8):
    worst, win, rx = 0, 0, 0
    for r in range(world):
        wide, narrow = hybrid_plan(n, r, world)
        nz = n // world
        worst = max(worst, sum(b1 - b0 for _, b0, b1 in wide) + nz * len(narrow))
        lo = min([max(0, r * nz - nz // 2)] + [b0 for _, b0, _ in wide]); hi = max([min(n, (r + 1) * nz + nz // 2)] + [b1 for _, _, b1 in wide])
        win = max(win, hi - lo)
        rx = max(rx, sum(narrow) * plane)
    tc = hyb[world]                                             # measured above (the plane-pass model would give worst * t1 / (n * passes))
    tx = rx / 150e9 * 1e3
    print("  %d   %8d           %5.2fx   %6d of %-6d    %8.2f                %8.2f     %8.2f .. %8.2f ms                          %.2fx .. %.2fx"
          % (world, worst, n * passes / worst, win, n, rx / 2**30, tc, max(tc, tx), tc + tx, t1 / (tc + tx), t1 / max(tc, tx)))
