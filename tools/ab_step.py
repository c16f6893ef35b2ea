"""Dev tool: A/B of the whole JFA (vp_jfa) per kernel between builds of libvphip.so, interleaved; prints per-kernel mean ms.
  python tools/ab_step.py --n 512 --libs a.so,b.so"""
import sys, os, math, argparse, ctypes, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import Frame
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512); ap.add_argument("--refine", type=int, default=24)
ap.add_argument("--libs", required=True); ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
_vp, _sz, fp = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Frame)
n = a.n
xyz, tri = M.bunny(a.refine); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
dev = torch.device("cuda", 0)
dx = torch.from_numpy(xyz.copy()).to(dev); dt = torch.from_numpy(tri.astype("int32")).to(dev)
g = torch.zeros(fr.words, dtype=torch.int32, device=dev); sdf = torch.empty(fr.voxels, dtype=torch.float32, device=dev)
libs = []
work = None                                                   # ONE workspace for every build (n = 2048: 129 GiB each otherwise)
for p in a.libs.split(","):
    L = ctypes.CDLL(os.path.abspath(p)); ctx = _vp()
    L.vp_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]; L.vp_ctx_create(0, ctypes.byref(ctx))
    L.vp_voxelize.argtypes = [_vp, fp, _vp, _vp, _sz, _vp, _sz, ctypes.c_int, ctypes.c_int]
    L.vp_jfa.argtypes = [_vp, fp, _vp, ctypes.c_float, _vp, _vp, _sz, ctypes.c_int]
    L.vp_prof_get.argtypes = [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.vp_prof_name.restype = ctypes.c_char_p; L.vp_prof_name.argtypes = [ctypes.c_int]
    L.vp_prof_enable.argtypes = [_vp, ctypes.c_int]; L.vp_prof_reset.argtypes = [_vp]; L.vp_ctx_sync.argtypes = [_vp]
    if work is None:
        L.vp_jfa_workspace_bytes.restype = _sz; L.vp_jfa_workspace_bytes.argtypes = [fp]
        work = torch.empty(L.vp_jfa_workspace_bytes(fr), dtype=torch.uint8, device=dev)
    libs.append((os.path.basename(p).replace("libvphip_", "").replace(".so", ""), L, ctx))
res = {nm: {} for nm, _, _ in libs}
chk = {}
for r in range(a.rounds + 1):
    order = libs if r == 0 else libs[r % len(libs):] + libs[:r % len(libs)]
    for nm, L, ctx in order:
        L.vp_prof_reset(ctx); L.vp_prof_enable(ctx, 1)
        for _ in range(2):
            L.vp_voxelize(ctx, fr, g.data_ptr(), dx.data_ptr(), dx.shape[0], dt.data_ptr(), dt.shape[0], 2, 0)
            L.vp_jfa(ctx, fr, g.data_ptr(), -math.inf, sdf.data_ptr(), work.data_ptr(), work.numel(), 2)
        L.vp_prof_enable(ctx, 0)
        if r == 0:
            L.vp_ctx_sync(ctx); chk[nm] = sum(int(c.to(torch.int64).sum().item()) for c in sdf.view(torch.int32).split(1 << 28)); continue   # in chunks: 8.6 G voxels at n = 2048
        for kern in range(20):
            ms, cnt = ctypes.c_double(), ctypes.c_uint64()
            if L.vp_prof_get(ctx, kern, ctypes.byref(ms), ctypes.byref(cnt)) == 0 and cnt.value:
                res[nm].setdefault(L.vp_prof_name(kern).decode(), []).append(ms.value / 2)
print("results identical:", len(set(chk.values())) == 1)
keys = sorted({k for v in res.values() for k in v})
print("%-12s" % "kernel" + "".join("%12s" % nm for nm, _, _ in libs) + "   (median ms per step)")
for k in keys:
    print("%-12s" % k + "".join("%12.4f" % statistics.median(res[nm].get(k, [0])) for nm, _, _ in libs))
print("%-12s" % "sum" + "".join("%12.4f" % sum(statistics.median(v) for v in res[nm].values()) for nm, _, _ in libs))
