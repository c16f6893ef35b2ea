"""Dev tool: A/B of the whole step (vp_voxelize + vp_jfa) by WALL CLOCK between builds of libvphip.so, interleaved, no per-kernel
events (what bench.py's `value` sees: launches, fills and gaps included).
  python tools/ab_wall.py --n 512 --libs a.so,b.so"""
import sys, os, math, argparse, ctypes, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import Frame
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512); ap.add_argument("--refine", type=int, default=24)
ap.add_argument("--libs", required=True); ap.add_argument("--rounds", type=int, default=9); ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
_vp, _sz, fp = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Frame)
n = a.n
xyz, tri = M.bunny(a.refine); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
dev = torch.device("cuda", 0)
dx = torch.from_numpy(xyz.copy()).to(dev); dt = torch.from_numpy(tri.astype("int32")).to(dev)
g = torch.zeros(fr.words, dtype=torch.int32, device=dev); sdf = torch.empty(fr.voxels, dtype=torch.float32, device=dev)
libs = []
for p in a.libs.split(","):
    L = ctypes.CDLL(os.path.abspath(p)); ctx = _vp()
    L.vp_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]; L.vp_ctx_create(0, ctypes.byref(ctx))
    L.vp_voxelize.argtypes = [_vp, fp, _vp, _vp, _sz, _vp, _sz, ctypes.c_int, ctypes.c_int]
    L.vp_jfa.argtypes = [_vp, fp, _vp, ctypes.c_float, _vp, _vp, _sz, ctypes.c_int]; L.vp_ctx_sync.argtypes = [_vp]
    libs.append((os.path.basename(p).replace("libvphip_", "").replace(".so", ""), L, ctx))
res = {nm: [] for nm, _, _ in libs}
for r in range(a.rounds + 1):
    order = libs[r % len(libs):] + libs[:r % len(libs)]
    for nm, L, ctx in order:
        L.vp_ctx_sync(ctx); t0 = time.perf_counter()
        for _ in range(a.steps):
            L.vp_voxelize(ctx, fr, g.data_ptr(), dx.data_ptr(), dx.shape[0], dt.data_ptr(), dt.shape[0], 2, 0)
            L.vp_jfa(ctx, fr, g.data_ptr(), -math.inf, sdf.data_ptr(), None, 0, 2)
        L.vp_ctx_sync(ctx); t1 = time.perf_counter()
        if r: res[nm].append((t1 - t0) / a.steps * 1e3)
for nm in res:
    print("%-10s median %.4f ms  min %.4f  max %.4f" % (nm, statistics.median(res[nm]), min(res[nm]), max(res[nm])))
