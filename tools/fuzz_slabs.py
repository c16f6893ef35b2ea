"""Dev tool (GPU box): randomized campaign for the Z-slab driver (vp_multi_*, several contexts on device 0): random grid, slab count,
transport (halo / ghost / hybrid / transpose) and algorithm; the concatenated slabs against the single-context vp_jfa, bit for bit.
  python tools/fuzz_slabs.py [--seconds 600] [--seed0 5000]"""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_mesh_voxelization_amd import capi
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, MULTI_GHOST, MULTI_HALO, MULTI_HYBRID, MULTI_TRANSPOSE, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=600.0); ap.add_argument("--seed0", type=int, default=5000)
a = ap.parse_args()
eng = Engine(0)
t_end, seed, done = time.time() + a.seconds, a.seed0, 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n = int(rng.choice([64, 128, 192, 256, 320, 384, 512]))
    world = int(rng.choice([g for g in (2, 3, 4, 5, 6, 8) if n % g == 0 and (n // g) % 8 == 0]))
    mode = int(rng.choice([MULTI_HALO, MULTI_GHOST, MULTI_HYBRID, MULTI_TRANSPOSE, MULTI_TRANSPOSE]))
    algo = ALGO_TILED if rng.random() < 0.8 else ALGO_NAIVE
    fill = -math.inf if rng.random() < 0.7 else math.inf
    vs = float(np.float32(10.0 ** rng.uniform(-2.5, 0.5)))
    origin = tuple(float(v) for v in ((rng.random(3) - 0.5) * 20.0).astype(np.float32))
    fr = Frame.make(n, vs, origin)
    kind = str(rng.choice(["noise", "sparse", "boxes", "slab-local"]))
    nw = fr.words
    if kind == "noise":
        words = rng.integers(0, 2**32, nw, dtype=np.uint32) & rng.integers(0, 2**32, nw, dtype=np.uint32)
    elif kind == "sparse":
        words = (rng.random(nw) < 10.0 ** rng.uniform(-5, -2)).astype(np.uint32) << rng.integers(0, 32, nw).astype(np.uint32)
    else:
        occ = np.zeros((n, n, n), bool)
        for _ in range(int(rng.integers(1, 8))):
            lo = rng.integers(0, n, 3); ext = np.where(rng.random(3) < 0.4, n, rng.integers(1, max(2, n // 3), 3))
            if kind == "slab-local":                              # everything inside ONE slab: the others see seeds only through the exchange
                r = int(rng.integers(0, world)); lo[2] = r * (n // world) + rng.integers(0, n // world); ext[2] = 1 + rng.integers(0, 4)
            hi = np.minimum(n, lo + ext); lo = np.where(ext == n, 0, lo)
            occ[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] ^= True
        words = np.packbits(occ.reshape(-1), bitorder="little").view(np.uint32)
    ref = eng.jfa(fr, eng.to_device(words, np.uint32), fill=fill, algo=algo).cpu().numpy()
    m = capi.Multi([0] * world)
    try:
        m.set_grid(fr, words)
        m.jfa(fill=fill, algo=algo, mode=mode)
        got = m.get_sdf()
    finally:
        m.close()
    ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    print("seed %d n=%d slabs=%d mode=%d algo=%d fill=%s %-10s %s" % (seed, n, world, mode, algo, fill, kind, "ok" if ok else "MISMATCH %d" % int((got.view(np.uint32) != ref.view(np.uint32)).sum())), flush=True)
    if not ok:
        sys.exit(1)
    done += 1; seed += 1
print("cases: %d mismatches: 0" % done)
