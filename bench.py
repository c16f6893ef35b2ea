#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s of (voxelize + JFA) at n = 512 on the 1,348,128-face bunny.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n {512,1024,2048}]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = tiled voxelization of the resident mesh into a bit-packed n^3 grid followed by the full JFA
(border mask, first pass from the mask, sparse / dense tile passes, last pass fused with the id -> sdf
conversion) into a float sdf -- both through the C ABI of libvphip.so, inputs already in HBM.
Rank 0 prints ONE JSON line (DESIGN.md "Measurement"):

  value / ms_per_step   whole job, wall clock between barriers, max over ranks
  roofline              the DOMINANT kernel only: the dense tile pass (jfa_dense), its own algorithmic bytes
                        (2 * S * n^3) / its own mean launch time from hipEvents on the kernel's stream
  kernels               every kernel of the step: launches per step, mean ms, algorithmic bytes, GB/s, fraction of peak
  n1024                 (N = 1, default size only) the same JFA at n = 1024, where the north star puts its roofline target
  cpu_baseline          the oracle (C restatement of the reference's sequential path) on the box's host cores:
                        1-thread voxelize + OpenMP JFA with ALL passes -- measured, not extrapolated
  copy_peak             (N = 1) vp_stream_copy over 1 GiB on this box: roofline.measured_copy_GBs / frac_of_measured
  config3               (N = 1, default size) BASELINE config 3 as a pipeline: bimba + bunny voxelize, CSG union, JFA at n = 512,
                        device-resident; kernels.csg_words against 3 n^3/8 bytes; checked against the reference's golden row
  vox_large_triangles   (N = 1, default size) the LDS tile rasteriser on a mesh of large triangles (vox_scan / vox_scatter / vox_tile per step),
                        checked against the naive voxelizer
  totals_incl_transfers (N = 1, default size) one host-in / host-out round (vp_voxelize_host + vp_jfa_host) = what the
                        reference's Compute() calls time as Memory + Processing (38.6 + 829.6 ms)

N > 1: strong scaling of the same n^3 job over N Z-slabs, one process per GPU (cuda_mesh_voxelization_amd/slab.py).  After the
timed region every rank runs the ONE-GPU path on its own device and compares its slab of the bitmask and of the sdf bit for
bit (`parity_ok`; a mismatch on any rank makes the run exit non-zero), then times the OTHER pipelines over shorter regions, checked
the same way: `multi_alt` = the transposed pipeline (cyclic planes, ONE RCCL all-to-all), `multi_alt_halo` = RCCL point-to-point halos
before every pass, `multi_alt_transpose_p2p` = the transposed pipeline with its planes sent point to point and placed directly -- under a
watchdog (VP_BENCH_ALT_TIMEOUT, default 180 s each): if a transport hangs, rank 0 prints the line of the timed pipeline as it stands, with the time-out recorded in that object,
and the job ends with exit code 2 (VP_BENCH_LENIENT=1: 0).
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
# the pool's host driver only supports dmabuf IPC: without this RCCL's device-memory sharing between the ranks of one node fails with
# `hipIpcGetMemHandle: invalid argument`.  Exported on the boxes already; kept here for any environment that launches this file bare.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")



def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--grid-n", dest="n", type=int, default=512, choices=[256, 512, 1024, 2048],
                    help="grid side; 512 = the headline configuration, 1024 / 2048 = the sizes the north star shards.  Under "
                         "torch.distributed.run spell it --grid-n: the launcher's own parser rejects a bare --n as an ambiguous "
                         "abbreviation of its --nnodes / --nproc-per-node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-n1024", action="store_true", help="skip the extra n = 1024 JFA block of the default run")
    ap.add_argument("--no-config3", action="store_true", help="skip the BASELINE config 3 block (bimba + bunny, CSG union, JFA at n = 512)")
    ap.add_argument("--no-host-totals", action="store_true", help="skip the host-in / host-out round (reference-style totals)")
    ap.add_argument("--no-copy-peak", action="store_true", help="skip the 1-GiB stream-copy measurement")
    ap.add_argument("--multi", choices=["ghost", "halo", "hybrid", "transpose", "transpose-p2p"], default="ghost",
                    help="N > 1: 'ghost' = Z-slabs with recomputed ghost planes, no data-path exchange (default: the one pipeline that needs no "
                         "transport the build could never try on hardware); 'transpose' = planes dealt cyclically for every pass whose step is a "
                         "multiple of N (no exchange, no ghost planes), ONE RCCL all-to-all, Z-slabs for the last log2 N passes ('transpose-p2p': the same "
                         "planes as one batch of point-to-point messages placed directly: no send / staging buffer, no weave); 'halo' = Z-slabs "
                         "with RCCL point-to-point halo planes before every pass; 'hybrid' = ghost planes for the wide passes (k > nz/2), halos of "
                         "the adjacent ranks -- sent a pass ahead, under the interior planes -- for the narrow ones")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment): this process touches no GPU -- torch is not even
    imported yet -- and starts the N ranks as a FRESH child (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
    <same arguments>`, never an exec), forwards rank 0's JSON line to its own stdout, everything else the ranks print to stderr, and
    returns the child's exit code.  The reference hard-wires device 0 (apps/cli/main.cpp:22-23); this is what replaces it for N > 1."""
    import subprocess
    argv = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--grid-n", str(args.n), "--multi", args.multi]
    for flag in ("no_cpu_baseline", "no_n1024", "no_config3", "no_host_totals", "no_copy_peak"):
        if getattr(args, flag):
            argv.append("--" + flag.replace("_", "-"))
    # --standalone: the launcher's own rendezvous on a port IT picks and keeps (binding one here and closing it again could lose it to another
    # process in between)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            lines += 1
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write("bench.py: the ranks printed %d JSON lines, expected 1\n" % lines)
        rc = 1
    return rc


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _early = parse_args()
    if _early.gpus > 1:
        sys.exit(launch_ranks(_early))

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_GRID = 512
REFINE = 24                      # 56,172 * 24 = 1,348,128 faces (benchmarks_v2/bunny_1348128)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0      # same guide: float4 copy, 79 % of spec
BASELINE_MVOX = 480.0            # reference tiled vox+JFA, n=512, kernels only (derived from its CSVs)


def kernel_bytes(n, planes, S, ntris, nverts):
    """Algorithmic bytes per launch (every array once per kernel that must touch it; S = id bytes)."""
    vox = n * n * planes
    return {
        "vox_setup": 12 * ntris + 12 * nverts, "vox_scan": 0, "vox_scatter": 0, "vox_tile": 0, "vox_naive": 12 * ntris + 12 * nverts,
        "vox_zero": vox // 8, "vox_fill": 2 * vox // 8, "csg_words": 3 * vox // 8,
        "surface": 2 * vox // 8, "jfa_init": vox // 8 + S * vox,
        "jfa_first": S * vox + vox // 8,                 # pure store stream + the border mask
        "jfa_sparse": 2 * S * vox, "jfa_dense": 2 * S * vox, "jfa_pass": 2 * S * vox,
        "jfa_last": S * vox + 4 * vox + vox // 8,        # ids in, floats out, bitmask in
        "jfa_final": S * vox + 4 * vox + vox // 8,
    }


def kernel_table(prof, steps, bytes_per):
    out = {}
    for k, v in prof.items():
        ms = v["ms"] / max(v["launches"], 1)
        b = bytes_per.get(k, 0)
        gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[k] = {"launches_per_step": round(v["launches"] / steps, 2), "avg_ms": round(ms, 4), "ms_per_step": round(v["ms"] / steps, 4),
                  "bytes": int(b), "GB/s": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def cpu_baseline(xyz, tri, origin, vs, n):
    """Oracle on the host cores, the whole job: sequential voxelize (1 thread = the reference's -t 0 semantics) and the
    JFA with every pass under OpenMP (the reference's -t 3).  ~10-30 s on the GPU box's host."""
    from oracle import oracle as O
    t0 = time.perf_counter()
    words = O.voxelize(xyz, tri, n, vs, origin)
    t_vox = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.jfa(words, n, vs, origin)
    t_jfa = time.perf_counter() - t0
    return {
        "value": round(n ** 3 / (t_vox + t_jfa) / 1e6, 3), "unit": "Mvoxels/s", "cores": O.threads(), "kind": "port",
        "voxelize_s_1thread": round(t_vox, 3), "jfa_s_openmp": round(t_jfa, 3),
        "sample": "n=%d bunny %d faces, the whole step: sequential voxelize on 1 thread (%.2f s) + JFA init and all %d passes with "
                  "OpenMP on %d threads (%.2f s); nothing extrapolated" % (n, tri.shape[0], t_vox, int(math.log2(n)), O.threads(), t_jfa),
    }


def copy_peak(eng, gib=1, reps=5):
    """HBM copy rate of THIS box: vp_stream_copy (16 B per lane, grid-stride, nothing computed) over `gib` GiB, bytes read +
    bytes written per second, best of `reps` (torch events: the context runs on torch's current stream)."""
    nbytes = gib << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
    src.zero_()
    for _ in range(2):
        eng.ctx.stream_copy(dst.data_ptr(), src.data_ptr(), nbytes)
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.ctx.stream_copy(dst.data_ptr(), src.data_ptr(), nbytes)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    del src, dst
    return {"GB/s": round(2 * nbytes / (best * 1e-3) / 1e9, 1), "bytes_copied": nbytes, "best_ms": round(best, 4), "reps": reps,
            "kernel": "vp_stream_copy (16 B per lane, grid-stride); rate = (bytes read + bytes written) / time"}


def popcount_words(t):
    """set bits of an int32 tensor (device), via a 256-entry byte table"""
    table = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=t.device)
    return int(table[t.view(torch.uint8).to(torch.int64)].sum().item())


def golden_row(meshes, n, op):
    """row of tests/golden/survey_table.json (outputs of the reference's own sequential path); None if absent"""
    try:
        rows = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_table.json")))["rows"]
    except Exception:
        return None
    for r in rows:
        if r["meshes"] == meshes and r["n"] == n and r["op"] == op:
            return r
    return None


def run_config3(eng, steps=10, warmup=2):
    """BASELINE config 3: bimba.obj then bunny.obj in the bbox frame of both (apps/cli/main.cpp:62-87), tiled voxelize each,
    CSG union into the first grid (-p 1), JFA sdf at n = 512 -- device-resident, through the same C-ABI calls."""
    from cuda_mesh_voxelization_amd import mesh as M
    from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
    n = 512
    a_xyz, a_tri = M.import_mesh(M.asset("bimba.obj"))
    b_xyz, b_tri = M.import_mesh(M.asset("bunny.obj"))
    origin, vs = M.frame([a_xyz, b_xyz], n)
    fr = Frame.make(n, vs, origin)
    da = eng.mesh_to_device(a_xyz, a_tri)
    db = eng.mesh_to_device(b_xyz, b_tri)
    ga, gb = eng.new_grid(fr), eng.new_grid(fr)
    sdf = torch.empty(fr.voxels, dtype=torch.float32, device=eng.device)

    def step():
        eng.voxelize(fr, da[0], da[1], out=ga, algo=ALGO_TILED)
        eng.voxelize(fr, db[0], db[1], out=gb, algo=ALGO_TILED)
        eng.csg(ga, gb, 1)                                          # VP_OP_UNION
        eng.jfa(fr, ga, out=sdf, algo=ALGO_TILED)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.ctx.prof_reset(); eng.ctx.prof_select(None); eng.ctx.prof_enable(True)
    for _ in range(TABLE_STEPS):
        step()
    torch.cuda.synchronize()
    eng.ctx.prof_enable(False)
    S = eng.ctx.jfa_id_bytes(fr)
    kb = kernel_bytes(n, n, S, int(a_tri.shape[0] + b_tri.shape[0]) // 2, int(a_xyz.shape[0] + b_xyz.shape[0]) // 2)
    kernels = kernel_table(eng.ctx.prof(), TABLE_STEPS, kb)
    row = golden_row(["bimba.obj", "bunny.obj"], n, 1)
    got = {"union_popcount": popcount_words(ga), "sdf_zeros": int((sdf == 0).sum().item()),
           "sdf_sum_pos": float(sdf[sdf > 0].double().sum().item()), "sdf_sum_neg": float(sdf[sdf < 0].double().sum().item())}
    ok = None
    if row:
        want = {"union_popcount": row["csg"][0], "sdf_zeros": row["sdf"]["zeros"], "sdf_sum_pos": row["sdf"]["sum_pos"], "sdf_sum_neg": row["sdf"]["sum_neg"]}
        # counts exact; the sums are double accumulations of identical floats in another order (tree vs index order): 1e-6 relative
        ok = (got["union_popcount"] == want["union_popcount"] and got["sdf_zeros"] == want["sdf_zeros"]
              and abs(got["sdf_sum_pos"] - want["sdf_sum_pos"]) <= 1e-6 * abs(want["sdf_sum_pos"])
              and abs(got["sdf_sum_neg"] - want["sdf_sum_neg"]) <= 1e-6 * abs(want["sdf_sum_neg"]))
    csg = kernels.get("csg_words", {})
    return {"workload": "bimba.obj (%d faces) + bunny.obj (%d faces), frame = bbox of both, tiled voxelize x2 + CSG union + JFA sdf at n = %d, "
                        "device-resident" % (a_tri.shape[0], b_tri.shape[0], n),
            "ms_per_step": round(elapsed / steps * 1e3, 4), "Mvoxels/s": round(n ** 3 / (elapsed / steps) / 1e6, 1), "steps": steps,
            "kernels": kernels, "kernels_ms_per_step": {k: v["ms_per_step"] for k, v in kernels.items()},
            "csg": {"avg_ms": csg.get("avg_ms"), "bytes": csg.get("bytes"), "GB/s": csg.get("GB/s"), "frac_of_peak": csg.get("frac_of_peak"),
                    "note": "3 n^3/8 = 48 MiB at n = 512: a ~10 us kernel, launch- and ramp-bound rather than HBM-bound; the reference's "
                            "kernel takes 1.59 ms (benchmarks_v2/bunny_1348128/bunny_1348128_naive_csg.csv:82-101)"},
            "checks": got, "golden": "tests/golden/survey_table.json (bimba + bunny, n = 512, union)", "parity_ok": ok}


def run_large_triangles(eng, steps=20, warmup=3):
    """The LDS tile rasteriser on its own: the headline mesh has no large triangle, so its voxelizer is vox_zero +
    vox_setup + vox_fill.  sphere.obj (1,280 faces) at n = 512 is the opposite case: every triangle spans dozens of 8 x 8-column tiles, all of
    them go through the record list, the tile histogram scan, the per-tile lists and the ds_xor tile kernel.  Checked bit for bit against the
    naive voxelizer (one thread per triangle, word-mask atomicXor) on the same device."""
    from cuda_mesh_voxelization_amd import mesh as M
    from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
    n = 512
    xyz, tri = M.import_mesh(M.asset("sphere.obj"))
    origin, vs = M.frame([xyz], n)
    fr = Frame.make(n, vs, origin)
    d = eng.mesh_to_device(xyz, tri)
    g, h = eng.new_grid(fr), eng.new_grid(fr)
    for _ in range(warmup):
        eng.voxelize(fr, d[0], d[1], out=g, algo=ALGO_TILED)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.voxelize(fr, d[0], d[1], out=g, algo=ALGO_TILED)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.ctx.prof_reset(); eng.ctx.prof_select(None); eng.ctx.prof_enable(True)
    for _ in range(TABLE_STEPS):
        eng.voxelize(fr, d[0], d[1], out=g, algo=ALGO_TILED)
    torch.cuda.synchronize()
    eng.ctx.prof_enable(False)
    table = eng.ctx.prof()
    eng.voxelize(fr, d[0], d[1], out=h, algo=ALGO_NAIVE)
    torch.cuda.synchronize()
    return {"workload": "sphere.obj (%d faces, every one of them large), tiled solid voxelize into a bit-packed %d^3 grid, device-resident" % (tri.shape[0], n),
            "ms_per_step": round(elapsed / steps * 1e3, 4), "Mvoxels/s": round(n ** 3 / (elapsed / steps) / 1e6, 1), "steps": steps,
            "kernels_ms_per_step": {k: round(v["ms"] / TABLE_STEPS, 4) for k, v in table.items()},
            "set_voxels": popcount_words(g), "equal_to_naive_voxelizer": bool(torch.equal(g, h)),
            "reference_ms": {"tiled_processing_kernel_n512_1.35M_faces": 2.0, "source": "SURVEY.md 8(a) a-10 (benchmarks_v2, unstated NVIDIA GPU); a different mesh: for scale only"}}


def host_totals(eng, frame, xyz, tri, algo):
    """Reference-style totals: host arrays in, host arrays out, synchronous -- vp_voxelize_host + vp_jfa_host,
    i.e. the Memory + Processing scopes of TiledVox / TiledJFA (vox/tiled.cu:504-575, jfa/tiled.cu:254-257,334-335).
    One warm round first (workspace growth), then one timed round."""
    words = np.zeros(frame.words, dtype=np.uint32)
    sdf = np.empty(frame.voxels, dtype=np.float32)
    xyz32 = np.ascontiguousarray(xyz, dtype=np.float32)
    tri32 = np.ascontiguousarray(tri, dtype=np.uint32)
    out = {}
    for rnd in ("warm", "timed"):
        words[:] = 0
        t0 = time.perf_counter()
        eng.ctx.voxelize_host(frame, words, xyz32, tri32, algo)
        t1 = time.perf_counter()
        sdf.fill(-np.inf)
        t2 = time.perf_counter()
        eng.ctx.jfa_host(frame, words, -math.inf, sdf, algo)
        t3 = time.perf_counter()
        out = {"voxelize_ms": round((t1 - t0) * 1e3, 3), "jfa_ms": round((t3 - t2) * 1e3, 3)}
    out["total_ms"] = round(out["voxelize_ms"] + out["jfa_ms"], 3)
    out["Mvoxels/s"] = round(frame.n ** 3 / (out["total_ms"] * 1e-3) / 1e6, 1)
    out["bytes_moved"] = {"h2d": int(xyz32.nbytes + tri32.nbytes + words.nbytes), "d2h": int(words.nbytes + sdf.nbytes)}
    out["reference_ms"] = {"voxelize_total": 38.6, "jfa_total": 829.6, "source": "BASELINE.md (benchmarks_v2/bunny_1348128 tiled_vox / tiled_jfa, n = 512, unstated NVIDIA GPU)"}
    out["note"] = "pageable host arrays over PCIe, synchronous calls; never part of `value`"
    del words, sdf
    return out


def profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


LIMITER = ("co-limited: vector-ALU issue (valu_issue_frac of the kernel's cycles) and fabric traffic (`traffic` / bytes_per_launch x "
           "the algorithmic bytes), profiles/jfa_dense_traffic.json; DESIGN.md section 4")
DOMINANT = "jfa_dense"
TABLE_STEPS = 5


def measure(eng, step, steps, warmup, barrier):
    """The timed region: `steps` steps between barriers, hipEvents only around the dominant kernel (an event pair costs ~3 us of
    stream time; bracketing all 18 launches of a step cost 0.11 ms = 3 % of it, tools/prof_overhead.py).  The per-kernel table
    of the other kernels comes from TABLE_STEPS further steps with every launch bracketed, outside the timed region."""
    for _ in range(warmup):
        step()
    barrier()
    eng.ctx.prof_reset()
    eng.ctx.prof_select([DOMINANT])
    eng.ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.ctx.prof_enable(False)
    live = eng.ctx.prof()
    eng.ctx.prof_reset()
    eng.ctx.prof_select(None)
    eng.ctx.prof_enable(True)
    for _ in range(TABLE_STEPS):
        step()
    barrier()
    eng.ctx.prof_enable(False)
    table = eng.ctx.prof()
    return elapsed, live, table


def merge_tables(live, steps, table, bytes_per):
    """kernel table: the dominant kernel from the timed region, the rest from the extra fully bracketed steps"""
    kernels = kernel_table(table, TABLE_STEPS, bytes_per)
    kernels.update(kernel_table(live, steps, bytes_per))
    return kernels


def run_single(eng, frame, d_xyz, d_tri, steps, warmup, algo):
    grid = eng.new_grid(frame)
    sdf = torch.empty(frame.voxels, dtype=torch.float32, device=eng.device)

    def step():
        eng.voxelize(frame, d_xyz, d_tri, out=grid, algo=algo)
        eng.jfa(frame, grid, out=sdf, algo=algo)

    return measure(eng, step, steps, warmup, torch.cuda.synchronize)


def main():
    args = parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:                                  # (a bare `--gpus N` never gets here: launch_ranks above)
        sys.exit("bench.py --gpus %d inside a job of WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU fallback"

    from cuda_mesh_voxelization_amd import mesh as M
    from cuda_mesh_voxelization_amd.capi import ALGO_TILED, Frame
    from cuda_mesh_voxelization_amd.pipeline import Engine

    # N > 1: libraries write to stdout on their own -- RCCL prints its version banner there with plain printf under NCCL_DEBUG=VERSION (set on
    # the pool's boxes), C-buffered, i.e. AFTER everything this process prints.  The one JSON line is what stdout is for: file descriptor 1
    # is pointed at stderr for the life of the rank and the line goes out through a duplicate of the original descriptor.
    json_fd = None
    if world > 1:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

    def emit(obj):
        line = json.dumps(obj)
        if json_fd is None:
            print(line, flush=True)
        else:
            os.write(json_fd, (line + "\n").encode())

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("VP_BENCH_SHARE_GPU") == "1" and torch.cuda.device_count() < world:
            # test rig only (tests/test_slab_gpu.py on a one-GPU box): the ranks share the GPUs there are and rendezvous over gloo,
            # since RCCL refuses two ranks on one device.  The numbers of such a run mean nothing; the code path is the driver's.
            local_rank %= torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n = args.n
    refine = REFINE if n <= 1024 else 192                  # n = 2048: the 10,785,024-face mesh of BASELINE config 5
    xyz, tri = M.bunny(refine)
    origin, vs = M.frame([xyz], n)
    frame = Frame.make(n, vs, origin)
    eng = Engine(local_rank)
    d_xyz, d_tri = eng.mesh_to_device(xyz, tri)
    # S of SURVEY.md 8(d), "as implemented": what a pass really streams per voxel -- 4 bytes; above n = 1024 the 5 bytes of the id windows,
    # on one GPU and in every slab pipeline (memory and wire)
    S = eng.ctx.jfa_state_bytes(frame, ALGO_TILED)
    passes = int(math.log2(n))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pipe_report, pipe_desc, parity, multi_alt, alts_ok = None, None, None, None, True
    if world == 1:
        barrier()
        elapsed, live, table = run_single(eng, frame, d_xyz, d_tri, args.steps, args.warmup, ALGO_TILED)
        planes = n
    else:
        from cuda_mesh_voxelization_amd.slab import HostStagedDist, make_pipeline
        p2p = HostStagedDist(dist) if dist.get_backend() == "gloo" else dist        # gloo (the shared-GPU test rig) moves CPU tensors only
        pw, pv = n * n // 32, n * n

        def run_pipeline(kind, steps, warmup):
            """time `steps` steps of one slab pipeline; returns (elapsed max over ranks, timers, report, this rank's slab of the
            bitmask and of the sdf as copies) and frees the pipeline's buffers (n = 2048: two id volumes are 128 GiB)"""
            pipe = make_pipeline(kind, eng, frame, rank, world, p2p)

            def step():
                pipe.voxelize(d_xyz, d_tri, algo=ALGO_TILED)
                pipe.jfa()

            el, lv, tb = measure(eng, step, steps, warmup, barrier)
            t = torch.tensor([el], dtype=torch.float64, device=eng.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = pipe.words if pipe.words.numel() == pipe.frame.words else pipe.words[pipe.z0 * pw:pipe.z1 * pw]
            got = (w.clone(), pipe.sdf.clone(), pipe.z0, pipe.z1)
            rep = dict(pipe.report(), describe=pipe.describe(), regions=getattr(pipe, "regions", None))
            del pipe, step
            torch.cuda.empty_cache()
            return float(t.item()), lv, tb, rep, got

        def same_bits(got, ref_words, ref_sdf):
            w, s_, z0, z1 = got
            return [bool(torch.equal(w, ref_words[z0 * pw:z1 * pw])),
                    bool(torch.equal(s_.view(torch.int32), ref_sdf[z0 * pv:z1 * pv].view(torch.int32)))]

        elapsed, live, table, pipe_report, got = run_pipeline(args.multi, args.steps, args.warmup)
        pipe_desc = pipe_report.pop("describe")
        planes = n // world
        regs = pipe_report.pop("regions")
        if regs:                                           # ghost planes: a dense pass covers the slab widened by the later steps
            dense = [b1 - b0 for k, b0, b1 in regs if k * 4 < n and k > 1]
            if dense:
                planes = sum(dense) / len(dense)
        # ---- self-check: the ONE-GPU path on this rank's own device, this rank's slab compared bit for bit.  First contact of the
        # transports (RCCL point-to-point, peer copies) with real hardware happens on boxes the build never sees: a number printed
        # for a wrong result would be worse than no number.
        ref_words = eng.voxelize(frame, d_xyz, d_tri, algo=ALGO_TILED)
        ref_sdf = eng.jfa(frame, ref_words, algo=ALGO_TILED)
        torch.cuda.synchronize()
        eng._work = None
        torch.cuda.empty_cache()
        flags = same_bits(got, ref_words, ref_sdf)
        del got
        # the timed pipeline's parity is gathered BEFORE the other transport is touched: whatever happens there, this stands
        fl0 = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=eng.device)
        all0 = [torch.zeros_like(fl0) for _ in range(world)]
        dist.all_gather(all0, fl0)
        per_rank0 = [[bool(v) for v in t.tolist()] for t in all0]
        parity = {"parity_ok": all(all(r) for r in per_rank0),
                  "per_rank": [{"rank": i, "bitmask_slab_equal": r[0], "sdf_slab_equal": r[1]} for i, r in enumerate(per_rank0)],
                  "against": "the one-GPU path (vp_voxelize + vp_jfa of the whole grid) run on each rank's own device after the timed region"}
        # ---- the other transport in the same job, over a shorter region (so that a scaling run shows RCCL moving halos and not
        # only barriers when the default is ghost planes, and the exchange-free figure when it is not)
        # `multi_alt` = the transposed pipeline (one all-to-all), then RCCL halos, then the transposed pipeline with its planes sent point to
        # point -- each of them a transport the build could never try on hardware, each verified against the one-GPU result and time-boxed
        alt_kinds = [k for k in ("transpose", "halo", "transpose-p2p", "ghost") if k != args.multi][:3]
        alt_keys = ["multi_alt"] + ["multi_alt_" + k.replace("-", "_") for k in alt_kinds[1:]]
        alt_steps = max(2, args.steps // 4)

        def alt_region(alt_kind):
            """every rank: time another pipeline, compare its slab, agree on the outcome; returns its `multi_alt*` object"""
            alt_error, aflags = None, [False, False]
            try:
                # the secondary measurement must not cost the primary one: an exception here (first contact of a transport with real
                # hardware) is reported in the line, the timed pipeline's figures and its parity check stand
                a_el, _a_live, a_table, a_rep, a_got = run_pipeline(alt_kind, alt_steps, 1)
                aflags = same_bits(a_got, ref_words, ref_sdf)
                del a_got
            except Exception as e:                                   # noqa: BLE001 -- reported, not swallowed
                alt_error = "%s: %s" % (type(e).__name__, e)
                a_el, a_table, a_rep = float("nan"), {}, {"describe": alt_kind, "error": alt_error}
            fl = torch.tensor([1 if f else 0 for f in aflags], dtype=torch.int32, device=eng.device)
            allf = [torch.zeros_like(fl) for _ in range(world)]
            dist.all_gather(allf, fl)
            per_rank = [[bool(v) for v in t.tolist()] for t in allf]
            recv = torch.tensor([float(a_rep.get("bytes_received_total", 0))], dtype=torch.float64, device=eng.device)
            dist.all_reduce(recv, op=dist.ReduceOp.SUM)
            a_rep.pop("regions", None)
            any_err = torch.tensor([1 if alt_error else 0], dtype=torch.int32, device=eng.device)
            dist.all_reduce(any_err, op=dist.ReduceOp.MAX)
            alt_failed = bool(any_err.item())                        # on ANY rank: then the region's figures mean nothing
            return {"pipeline": alt_kind, "parallelism": a_rep.pop("describe"), "steps": alt_steps, "warmup": 1,
                    "ms_per_step": None if alt_failed else round(a_el / alt_steps * 1e3, 4),
                    "value": None if alt_failed else round(n ** 3 / (a_el / alt_steps) / 1e6, 2), "unit": "Mvoxels/s",
                    "bytes_received_per_step_all_ranks": int(recv.item() / (alt_steps + 1 + TABLE_STEPS)),
                    "error": alt_error if alt_error else ("another rank failed" if alt_failed else None),
                    "parity_ok": None if alt_failed else all(all(r) for r in per_rank),
                    "per_rank": [{"rank": i, "bitmask_slab_equal": r[0], "sdf_slab_equal": r[1]} for i, r in enumerate(per_rank)],
                    "report_rank0": a_rep,
                    "kernels_ms_per_step_rank0": {k: round(v["ms"] / TABLE_STEPS, 4) for k, v in a_table.items()}}

    peak = copy_peak(eng) if (world == 1 and not args.no_copy_peak) else None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n ** 3 / (elapsed / args.steps) / 1e6
        bytes_per = kernel_bytes(n, planes, S, int(tri.shape[0]), int(xyz.shape[0]))
        kernels = merge_tables(live, args.steps, table, bytes_per)
        dom = DOMINANT if DOMINANT in kernels else max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        kd = kernels[dom]
        tj = profile_json("jfa_dense_traffic.json") or {}
        fl = profile_json("formulation_floor.json") or {}
        traffic = tj.get("hbm_bytes_per_launch") if (n == N_GRID and world == 1) else None
        counters_from = ("profiles/jfa_dense_traffic.json (round %s: separate rocprofv3 --pmc passes on the builder's box, "
                         "NOT this run)" % tj.get("round", "?")) if traffic else None
        out = {
            "metric": "Mvoxels/s (voxelize+JFA) at N=%d, bunny %.2fM tris" % (n, tri.shape[0] / 1e6),
            "value": round(value, 2), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong",        # the same n^3 job for every N (N > 1 splits it into Z-slabs)
            "vs_baseline": round(value / BASELINE_MVOX, 2) if n == N_GRID else None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "bunny.obj refined x%d (%d faces, %d verts) -> tiled solid voxelize into bit-packed %d^3 grid + JFA sdf "
                                   "(border mask + %d passes, last one fused with id -> sdf), device-resident"
                                   % (refine, tri.shape[0], xyz.shape[0], n, passes),
                       "n": n, "triangles": int(tri.shape[0]), "jfa_state_bytes": S,
                       "parallelism": "1 gpu" if world == 1 else pipe_desc,
                       "world_size_seen": world,
                       "baseline": "480 Mvoxels/s = reference tiled vox+JFA kernels-only at n=512 (BASELINE.md, unstated NVIDIA GPU)"},
            # `bound` names the roofline the kernel is priced against (the contract: "hbm" | "mfma"); `limiter` says what the counters
            # show actually holds it back
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": kd["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kd["frac_of_peak"], "frac_of_achievable": round(kd["GB/s"] / HBM_ACHIEVABLE_GBS, 4),
                         "achievable": HBM_ACHIEVABLE_GBS, "achievable_source": "MI355X_MICROARCH.md (float4 copy); this box: measured_copy_GBs",
                         "measured_copy_GBs": peak["GB/s"] if peak else None,
                         "frac_of_measured": round(kd["GB/s"] / peak["GB/s"], 4) if peak else None,
                         "traffic": traffic, "traffic_source": counters_from,
                         "bytes_per_launch": kd["bytes"], "avg_launch_ms": kd["avg_ms"], "launches": int(round(kd["launches_per_step"] * args.steps)),
                         "valu_issue_frac": tj.get("valu_issue_frac") if traffic else None, "valu_issue_source": counters_from,
                         "limiter": LIMITER,
                         # what ANY exact evaluation of the 27-candidate scatter formulation could reach on this chip (tools/ubench/floor.hip:
                         # the irreducible instructions and the two streams, nothing else); measured on the builder's box, not in this run
                         "formulation_floor_frac": (fl.get("n%d" % n) or {}).get("formulation_floor_frac"),
                         "formulation_floor_source": fl.get("source") if fl.get("n%d" % n) else None,
                         # which round's files the quoted (not live) figures of this object come from
                         "quoted_from": {"profiles/formulation_floor.json": "round %s" % fl.get("round", "?"), "profiles/jfa_dense_traffic.json": "round %s" % tj.get("round", "?")},
                         "timing": "hipEvents on the kernel's stream around each of its launches inside the timed region; the other "
                                   "kernels of `kernels` are timed over %d further steps outside it" % TABLE_STEPS,
                         "note": "bytes = 2*S*n^2*planes (SURVEY.md 8(d)); 27 exact candidate evaluations per voxel (DESIGN.md section 4)"},
            "kernels": kernels,
            "kernels_ms_per_step": {k: v["ms_per_step"] for k, v in kernels.items()},
        }
        if peak:
            out["copy_peak"] = peak
        if pipe_report is not None:
            out["multi"] = pipe_report
            out["parity_ok"] = parity["parity_ok"]
            out["parity"] = parity
        if world == 1 and n == N_GRID and not args.no_n1024:
            # the north star's roofline target lives at n = 1024: same mesh, JFA only is what differs in cost per voxel
            n2 = 1024
            o2, v2 = M.frame([xyz], n2)
            f2 = Frame.make(n2, v2, o2)
            e2, l2, t2 = run_single(eng, f2, d_xyz, d_tri, 3, 1, ALGO_TILED)
            k2 = merge_tables(l2, 3, t2, kernel_bytes(n2, n2, eng.ctx.jfa_state_bytes(f2, ALGO_TILED), int(tri.shape[0]), int(xyz.shape[0])))
            jfa_ms = sum(v["ms_per_step"] for k, v in k2.items() if k.startswith("jfa_") or k == "surface")
            jfa_bytes = sum(v["bytes"] * v["launches_per_step"] for k, v in k2.items() if k.startswith("jfa_") or k == "surface")
            out["n1024"] = {"ms_per_step": round(e2 / 3 * 1e3, 3), "Mvoxels/s": round(n2 ** 3 / (e2 / 3) / 1e6, 1), "jfa_ms": round(jfa_ms, 3),
                            "jfa_GB/s": round(jfa_bytes / (jfa_ms * 1e-3) / 1e9, 1), "jfa_frac_of_peak": round(jfa_bytes / (jfa_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "dense_pass_ms": k2.get("jfa_dense", {}).get("avg_ms"), "dense_frac_of_peak": k2.get("jfa_dense", {}).get("frac_of_peak"),
                            "kernels": k2}
            # where the north star puts its bar ("70 % of HBM roofline on JFA at N = 1024"): the same object as `roofline`, at n = 1024
            kd2 = k2.get("jfa_dense", {})
            t2 = (tj.get("n1024") or {})
            out["roofline_n1024"] = {"kernel": "jfa_dense", "bound": "hbm", "achieved": kd2.get("GB/s"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": kd2.get("frac_of_peak"), "bytes_per_launch": kd2.get("bytes"), "avg_launch_ms": kd2.get("avg_ms"),
                                     "traffic": t2.get("hbm_bytes_per_launch"), "valu_issue_frac": t2.get("valu_issue_frac"), "limiter": LIMITER,
                                     "measured_copy_GBs": peak["GB/s"] if peak else None,
                                     "frac_of_measured": round(kd2.get("GB/s", 0.0) / peak["GB/s"], 4) if peak and kd2.get("GB/s") else None,
                                     "traffic_source": counters_from, "valu_issue_source": counters_from,
                                     "jfa_all_passes_frac": out["n1024"]["jfa_frac_of_peak"], "target_frac": 0.70,
                                     "formulation_floor_frac": (fl.get("n1024") or {}).get("formulation_floor_frac"),
                                     "distance_only_floor_frac": (fl.get("n1024") or {}).get("distance_only_floor_frac"),
                                     "formulation_floor_source": fl.get("source"),
                                     "timing": "hipEvents on the kernel's stream, 3 steps after the timed region of the headline workload"}
            # the number the north star targets, inside the object the driver parses: the dense pass at n = 1024 beside the headline's
            out["roofline"]["frac_n1024"] = kd2.get("frac_of_peak")
            out["roofline"]["n1024"] = {"achieved": kd2.get("GB/s"), "avg_launch_ms": kd2.get("avg_ms"), "bytes_per_launch": kd2.get("bytes"),
                                        "jfa_all_passes_frac": out["n1024"]["jfa_frac_of_peak"], "target_frac": 0.70,
                                        "formulation_floor_frac": (fl.get("n1024") or {}).get("formulation_floor_frac")}
        if world == 1 and n == N_GRID and not args.no_config3:
            out["config3"] = run_config3(eng)
            if "csg_words" in out["config3"]["kernels"]:
                out["kernels"]["csg_words"] = dict(out["config3"]["kernels"]["csg_words"], measured_in="config3 (the headline step has no CSG)")
        if world == 1 and n == N_GRID and not args.no_config3:
            out["vox_large_triangles"] = run_large_triangles(eng)
        if world == 1 and n == N_GRID and not args.no_host_totals:
            out["totals_incl_transfers"] = host_totals(eng, frame, xyz, tri, ALGO_TILED)
            out["totals_incl_transfers_ms"] = out["totals_incl_transfers"]["total_ms"]
        if world == 1 and not args.no_cpu_baseline and n == N_GRID:
            out["cpu_baseline"] = cpu_baseline(xyz, tri, origin, vs, n)
    if world > 1:
        # The other pipelines run under a watchdog: a transport that HANGS on hardware the build never saw (a collective that never completes
        # cannot be cancelled) must not take the timed pipeline's line with it.  After VP_BENCH_ALT_TIMEOUT seconds (default 180, per
        # pipeline) rank 0 prints the line it already has, with the time-out recorded in that pipeline's object, and every rank leaves.
        import threading
        limit = float(os.environ.get("VP_BENCH_ALT_TIMEOUT", "180"))
        lock, finished, printed, current = threading.Lock(), [False], [False], [0]
        # exit code of a run in which another transport HUNG: 2 -- the line's own measurement and its parity check stand and the time-out is in
        # the line, but whoever only looks at the exit status must see that something never came back; VP_BENCH_LENIENT=1: 0
        hang_rc = 0 if os.environ.get("VP_BENCH_LENIENT") == "1" else 2
        alts = {}

        def give_up():
            with lock:
                if finished[0]:
                    return
                if rank == 0 and not printed[0]:
                    out.update(alts)
                    out[alt_keys[current[0]]] = {"pipeline": alt_kinds[current[0]], "steps": alt_steps, "ms_per_step": None, "value": None, "parity_ok": None,
                                                 "error": "no result within %.0f s (VP_BENCH_ALT_TIMEOUT): the transport hung; the timed "
                                                          "pipeline's figures and parity above stand" % limit}
                    emit(out)
                sys.stdout.flush()
                sys.stderr.write("bench.py: rank %d gave up on the %s pipeline after %.0f s\n" % (rank, alt_kinds[current[0]], limit))
                sys.stderr.flush()
                os._exit(hang_rc if parity["parity_ok"] else 1)

        # a timer stays armed until every rank is through the LAST barrier: a rank that finished the region while another one timed out
        # and left would otherwise wait in that barrier for ever
        for i, kind in enumerate(alt_kinds):
            with lock:
                current[0] = i
            timer = threading.Timer(limit, give_up)
            timer.daemon = True
            timer.start()
            alts[alt_keys[i]] = alt_region(kind)
            if i + 1 < len(alt_kinds):
                dist.barrier()
                timer.cancel()
        multi_alt = alts["multi_alt"]
        del ref_words, ref_sdf
        torch.cuda.empty_cache()
        alts_ok = all(a["parity_ok"] is not False for a in alts.values())       # a transport that RAN and disagreed fails the line
        with lock:
            if rank == 0:
                out.update(alts)
                out["parity_ok"] = parity["parity_ok"] and alts_ok
                emit(out)
            printed[0] = True
        dist.barrier()
        with lock:
            finished[0] = True
        timer.cancel()
        dist.destroy_process_group()
    elif rank == 0:
        emit(out)

    bad = parity is not None and not (parity["parity_ok"] and alts_ok)
    if bad:
        sys.exit("bench.py: a rank's slab differs from the one-GPU result (see `parity` / `multi_alt*` in the JSON line)")


if __name__ == "__main__":
    main()
