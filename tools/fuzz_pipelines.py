"""Dev tool (GPU box): randomized campaign for the Python Z-slab pipelines (slab.py: what `bench.py --gpus N` runs): random mesh (asset
under a random similarity, or a triangle soup), grid side, rank count and pipeline (ghost planes; RCCL-halo, hybrid and transposed
with ranks emulated by threads over the loopback of tests/test_slab_gpu.py), all on id windows, now and then above n = 1024 (5-byte layout); every rank's slab of the bitmask and of
the sdf against the whole-grid run, bit for bit.
  python tools/fuzz_pipelines.py [--seconds 600] [--seed0 20000]"""
import argparse, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_slab_gpu as T                                         # LoopbackDist, _run_slabs (emulated ranks)
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
from cuda_mesh_voxelization_amd.slab import GhostSlabPipeline, HipSlabBackend

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=600.0); ap.add_argument("--seed0", type=int, default=20000)
a = ap.parse_args()
eng = Engine(0)
assets = [M.import_mesh(M.asset(nm)) for nm in ("bunny.obj", "bimba.obj", "torus.obj", "sphere.obj", "d20.obj")]
t_end, seed, done = time.time() + a.seconds, a.seed0, 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n = int(rng.choice([96, 128, 192, 256, 288, 384, 512, 1152] if rng.random() < 0.9 else [1152, 1280]))
    world = int(rng.choice([g for g in (2, 3, 4, 6, 8) if n % g == 0 and (n // g) % 8 == 0]))
    kind = str(rng.choice(["ghost", "halo", "hybrid", "transpose", "transpose"]))     # (3 and 6 ranks: transpose runs its ghost fallback)
    algo = ALGO_TILED if (n > 512 or rng.random() < 0.8) else ALGO_NAIVE          # the voxelizer's and the reference run's algorithm
    if rng.random() < 0.6:                                        # an asset, rotated about z by a multiple of 90 degrees, scaled, moved
        xyz, tri = assets[int(rng.integers(len(assets)))]
        q = int(rng.integers(4)); c, s = [(1, 0), (0, 1), (-1, 0), (0, -1)][q]
        rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
        xyz = (xyz @ rot.T * np.float32(10.0 ** rng.uniform(-1, 1)) + ((rng.random(3) - 0.5) * 10).astype(np.float32)).astype(np.float32)
        origin, vs = M.frame([xyz], n)
        desc = "asset"
    else:                                                         # a soup in a free frame
        vs = float(np.float32(10.0 ** rng.uniform(-2, 0))); origin = ((rng.random(3) - 0.5) * 10).astype(np.float32)
        side = n * vs
        t = [origin + rng.random(3) * side + (rng.random((3, 3)) - 0.5) * side * 10.0 ** rng.uniform(-2.5, 0.0) for _ in range(int(rng.integers(20, 300)))]
        xyz = np.concatenate(t).astype(np.float32); tri = np.arange(xyz.shape[0], dtype=np.uint32).reshape(-1, 3)
        desc = "soup"
    fr = Frame.make(n, float(vs), tuple(float(v) for v in origin))
    dx, dt = eng.mesh_to_device(xyz, tri)
    ref_w = eng.voxelize(fr, dx, dt, algo=algo)
    ref_s = eng.jfa(fr, ref_w, algo=algo).clone()
    ok = True
    if kind.startswith("ghost"):
        nzv, pw = fr.voxels // world, n * n // 32
        for r in range(world):
            pipe = GhostSlabPipeline(HipSlabBackend(eng), fr, r, world)
            pipe.voxelize(dx, dt)
            s = pipe.jfa()
            w = pipe.words if pipe.words.numel() == fr.words else None
            ok &= bool(torch.equal(s.view(torch.int32), ref_s[r * nzv:(r + 1) * nzv].view(torch.int32)))
            if w is not None:
                ok &= bool(torch.equal(w[pipe.z0 * pw:pipe.z1 * pw], ref_w[pipe.z0 * pw:pipe.z1 * pw]))
            del pipe, s
    else:
        words, sdf = T._run_slabs(world, fr, xyz, tri, algo, kind=kind)
        ok = np.array_equal(words, eng.words_to_numpy(ref_w)) and np.array_equal(sdf.view(np.uint32), ref_s.cpu().numpy().view(np.uint32))
    print("seed %d n=%d ranks=%d %-12s algo=%d %-5s %s" % (seed, n, world, kind, algo, desc, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
    del ref_w, ref_s
    gc.collect(); torch.cuda.empty_cache()
    done += 1; seed += 1
print("cases: %d mismatches: 0" % done)
