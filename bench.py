#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s of (voxelize + JFA) at n = 512 on the 1,348,128-face bunny.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = tiled voxelization of the resident mesh into a bit-packed 512^3 grid followed by the
full JFA (init + 9 passes + finalize) into a float sdf -- both through the C ABI of libvphip.so,
inputs already in HBM.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").

N > 1: strong scaling of the same 512^3 job over N Z-slabs, one process per GPU.  Default: ghost planes are
recomputed instead of exchanged (a plane costs ~1.6 us to recompute and ~20 us to move over xGMI);
--multi halo selects the RCCL point-to-point halo exchange.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_GRID = 512
REFINE = 24                      # 56,172 * 24 = 1,348,128 faces (benchmarks_v2/bunny_1348128)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
STATE_BYTES = 4                  # JFA state per voxel as implemented (packed seed id)
BASELINE_MVOX = 480.0            # reference tiled vox+JFA, n=512, kernels only (derived from its CSVs)


def cpu_baseline(xyz, tri, origin, vs):
    """Oracle (C restatement of the reference's sequential path, OpenMP over the host cores) on a
    bounded sample of the same workload: the full voxelization, the JFA initialisation and the first
    two of the nine JFA passes at n = 512; pass time is extrapolated to nine passes."""
    from oracle import oracle as O
    n = N_GRID
    t0 = time.perf_counter()
    words = O.voxelize(xyz, tri, n, vs, origin)
    t_vox = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.jfa(words, n, vs, origin, max_passes=0)
    t_init = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.jfa(words, n, vs, origin, max_passes=2)
    t_two = time.perf_counter() - t0 - t_init
    passes = int(math.log2(n))
    est = t_vox + t_init + max(t_two, 0.0) * passes / 2.0
    return {
        "value": round(n ** 3 / est / 1e6, 3), "unit": "Mvoxels/s", "cores": O.threads(), "kind": "port",
        "sample": "n=512 bunny 1,348,128 faces: full sequential voxelize (%.2fs, 1 thread) + JFA init (%.2fs) + first 2 of %d "
                  "passes (%.2fs) with OpenMP; pass time extrapolated x%d/2" % (t_vox, t_init, passes, t_two, passes),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multi", choices=["ghost", "halo"], default="ghost",
                    help="N > 1: 'ghost' = communication-free Z-slabs with recomputed ghost planes (default); "
                         "'halo' = Z-slabs with RCCL point-to-point halo exchange between JFA passes")
    ap.add_argument("--n", type=int, default=N_GRID, help=argparse.SUPPRESS)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU fallback"

    from cuda_mesh_voxelization_amd import mesh as M
    from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
    from cuda_mesh_voxelization_amd.pipeline import Engine

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n = args.n
    xyz, tri = M.bunny(REFINE)
    origin, vs = M.frame([xyz], n)
    frame = Frame.make(n, vs, origin)
    eng = Engine(local_rank)
    d_xyz, d_tri = eng.mesh_to_device(xyz, tri)

    if world == 1:
        grid = eng.new_grid(frame)
        sdf = torch.empty(frame.voxels, dtype=torch.float32, device=eng.device)

        def step():
            eng.voxelize(frame, d_xyz, d_tri, out=grid, algo=ALGO_TILED)
            eng.jfa(frame, grid, out=sdf, algo=ALGO_TILED)
    else:
        from cuda_mesh_voxelization_amd.slab import GhostSlabPipeline, HipSlabBackend, SlabPipeline
        if args.multi == "ghost":
            pipe = GhostSlabPipeline(HipSlabBackend(eng), frame, rank, world)
        else:
            pipe = SlabPipeline(HipSlabBackend(eng), frame, rank, world, dist)

        def step():
            pipe.voxelize(d_xyz, d_tri, algo=ALGO_TILED)
            pipe.jfa(algo=ALGO_TILED)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.ctx.prof_reset()
    eng.ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.ctx.prof_enable(False)
    prof = eng.ctx.prof()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n ** 3 / (elapsed / args.steps) / 1e6
        passes = int(math.log2(n))
        kp = prof.get("jfa_pass", {"ms": 0.0, "launches": 0})
        # mean planes per jfa_pass launch on this rank (ghost mode widens the slab by the reach of later passes)
        planes = pipe.planes_computed / passes if (world > 1 and args.multi == "ghost") else n // world
        alg_bytes = int(2 * STATE_BYTES * n * n * planes)              # one id read + one id write per voxel
        avg_ms = kp["ms"] / max(kp["launches"], 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "jfa_pass_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mvoxels/s (voxelize+JFA) at N=512, bunny 1.35M tris",
            "value": round(value, 2), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong",        # the same 512^3 job for every N (N > 1 splits it into Z-slabs)
            "vs_baseline": round(value / BASELINE_MVOX, 2), "dtype": "f32", "data": "synthetic",
            "config": {"workload": "bunny.obj refined x24 (1,348,128 faces, 680k verts) -> tiled solid voxelize into bit-packed "
                                   "%d^3 grid + JFA sdf (init + %d passes + finalize), device-resident" % (n, passes),
                       "n": n, "triangles": int(tri.shape[0]), "jfa_state_bytes": STATE_BYTES,
                       "parallelism": "1 gpu" if world == 1 else ("z-slab x%d, ghost planes recomputed, no exchange" % world if args.multi == "ghost"
                                                                   else "z-slab x%d, RCCL p2p halo exchange" % world),
                       "baseline": "480 Mvoxels/s = reference tiled vox+JFA kernels-only at n=512 (BASELINE.md, unstated NVIDIA GPU)"},
            "roofline": {"kernel": "jfa_pass", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4), "launches": kp["launches"]},
            "kernels_ms_per_step": {k: round(v["ms"] / args.steps, 4) for k, v in prof.items()},
        }
        if world == 1 and not args.no_cpu_baseline and n == N_GRID:
            out["cpu_baseline"] = cpu_baseline(xyz, tri, origin, vs)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
