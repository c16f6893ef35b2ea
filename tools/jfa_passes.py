"""Dev tool: per-pass device time of the JFA kernels at a given n (hipEvent timing via vp_prof)."""
import sys, os, math, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine

ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=512); ap.add_argument("--refine", type=int, default=24)
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--algos", default="2")
a = ap.parse_args()
n = a.n
xyz, tri = M.bunny(a.refine); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.voxelize(fr, dx, dt)
idw = eng.ctx.jfa_id_bytes(fr) // 4                      # 32-bit words per id (2 above n = 1024)
ids = [torch.empty(fr.voxels * idw, dtype=torch.int32, device=eng.device) for _ in range(2)]
sdf = torch.empty(fr.voxels, dtype=torch.float32, device=eng.device)
ctx = eng.ctx
for algo in [int(x) for x in a.algos.split(",")]:
    # build the true per-pass inputs once, keep them (n<=512: 9 x 512 MiB fits easily)
    states = []
    ctx.jfa_init(fr, g.data_ptr(), None, None, ids[0].data_ptr())
    cur = ids[0].clone(); k = n // 2
    while k >= 1:
        nxt = torch.empty_like(cur)
        ctx.jfa_pass(fr, k, cur.data_ptr(), None, None, nxt.data_ptr(), algo)
        states.append((k, cur)); cur = nxt; k //= 2
    eng.sync()
    out = torch.empty_like(cur)
    tot = 0.0
    for k, st in states:
        none = -1 if idw == 2 else (0xFF9FF200 if n <= 512 else 0xFFDFFC00) - (1 << 32)   # jfa_common.h: Id64 / IdU<9> / IdU<10> as int32
        seeds = int((st[::idw] != none).sum().item())
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(a.reps):
            ctx.jfa_pass(fr, k, st.data_ptr(), None, None, out.data_ptr(), algo)
        ctx.prof_enable(False)
        pr = ctx.prof()                                            # the tile kernels are timed under one key per variant
        ms = sum(v["ms"] for v in pr.values()) / max(sum(v["launches"] for v in pr.values()), 1); tot += ms
        print("algo %d k=%4d seeded=%5.1f%%  %.3f ms  %.0f GB/s alg" % (algo, k, 100.0 * seeds / fr.voxels, ms, 8.0 * fr.voxels / ms / 1e6))
    print("algo %d total passes %.3f ms" % (algo, tot))
    # the fused last pass (k = 1 + id -> sdf) on the true k = 1 input
    k, st = states[-1]
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(a.reps):
        ctx.jfa_last_pass(fr, st.data_ptr(), None, None, out.data_ptr(), g.data_ptr(), -math.inf, sdf.data_ptr(), algo)
    ctx.prof_enable(False)
    pr = ctx.prof()
    print("algo %d fused last pass: %s" % (algo, {kk: round(v["ms"] / max(v["launches"], 1), 3) for kk, v in pr.items() if v["launches"]}))
