"""Dev tool: A/B timing of JFA pass kernels from several builds of libvphip.so in ONE process, interleaved round-robin
(single runs on this pool differ by several per cent between boxes and over time; medians over interleaved rounds do not).

  python tools/ab_pass.py --n 512 --k 4,1 --final --libs tools/exp/libvphip_a.so,tools/exp/libvphip_b.so

All libraries must share the id format (states are produced once, with the first one)."""
import sys, os, math, argparse, ctypes, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import Frame

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512); ap.add_argument("--refine", type=int, default=24)
ap.add_argument("--k", default="4"); ap.add_argument("--final", action="store_true")
ap.add_argument("--libs", required=True); ap.add_argument("--rounds", type=int, default=7); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
_vp, _sz, fp = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Frame)


class Lib:
    def __init__(self, path):
        self.name = os.path.basename(path).replace("libvphip_", "").replace(".so", "")
        L = self.L = ctypes.CDLL(os.path.abspath(path))
        L.vp_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]
        L.vp_jfa_pass.argtypes = [_vp, fp, ctypes.c_uint32, _vp, _vp, _vp, _vp, ctypes.c_int]
        L.vp_jfa_last_pass.argtypes = [_vp, fp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int]
        L.vp_jfa_init.argtypes = [_vp, fp, _vp, _vp, _vp, _vp]
        L.vp_voxelize.argtypes = [_vp, fp, _vp, _vp, _sz, _vp, _sz, ctypes.c_int, ctypes.c_int]
        L.vp_prof_enable.argtypes = [_vp, ctypes.c_int]; L.vp_prof_reset.argtypes = [_vp]
        L.vp_prof_get.argtypes = [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
        L.vp_ctx_sync.argtypes = [_vp]; L.vp_last_error.restype = ctypes.c_char_p
        self.ctx = _vp()
        self.ok(L.vp_ctx_create(0, ctypes.byref(self.ctx)))

    def ok(self, rc):
        if rc: raise RuntimeError("%s: %s" % (self.name, self.L.vp_last_error().decode()))

    def timed(self, fn, reps):
        L = self.L
        self.ok(L.vp_prof_reset(self.ctx)); self.ok(L.vp_prof_enable(self.ctx, 1))
        for _ in range(reps): self.ok(fn())
        self.ok(L.vp_prof_enable(self.ctx, 0))
        ms, cnt = ctypes.c_double(), ctypes.c_uint64()
        tot, n = 0.0, 0
        for kern in range(16):                                   # every jfa pass key (the enum grew over the rounds)
            if L.vp_prof_get(self.ctx, kern, ctypes.byref(ms), ctypes.byref(cnt)) == 0 and cnt.value:
                tot += ms.value; n += cnt.value
        return tot / max(n, 1)


libs = [Lib(p) for p in a.libs.split(",")]
n = a.n
xyz, tri = M.bunny(a.refine); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
dev = torch.device("cuda", 0)
dx = torch.from_numpy(xyz).to(dev); dt = torch.from_numpy(tri.astype("int32")).to(dev)
g = torch.zeros(fr.words, dtype=torch.int32, device=dev)
L0 = libs[0]
L0.ok(L0.L.vp_voxelize(L0.ctx, fr, g.data_ptr(), dx.data_ptr(), dx.shape[0], dt.data_ptr(), dt.shape[0], 2, 0))
idt = torch.int64 if n > 1024 else torch.int32                  # ids are 8 bytes above n = 1024
cur = torch.empty(fr.voxels, dtype=idt, device=dev)
L0.ok(L0.L.vp_jfa_init(L0.ctx, fr, g.data_ptr(), None, None, cur.data_ptr()))
want = sorted({int(x) for x in a.k.split(",")} | ({1} if a.final else set()), reverse=True)
states = {}
k = n // 2
while k >= 1:
    if k in want: states[k] = cur.clone() if len(want) > 1 else cur
    if k == min(want): break
    nxt = torch.empty_like(cur)
    L0.ok(L0.L.vp_jfa_pass(L0.ctx, fr, k, cur.data_ptr(), None, None, nxt.data_ptr(), 2))
    cur = nxt; k //= 2
L0.L.vp_ctx_sync(L0.ctx)
del cur
out = torch.empty(fr.voxels, dtype=idt, device=dev)
sdf = torch.empty(fr.voxels if a.final else 1, dtype=torch.float32, device=dev)
cases = [("k=%d" % k, k) for k in [int(x) for x in a.k.split(",")]] + ([("final", 0)] if a.final else [])
res = {(l.name, c): [] for l in libs for c, _ in cases}
ref = {}
for r in range(a.rounds + 1):
    order = libs if r == 0 else libs[r % len(libs):] + libs[:r % len(libs)]      # rotate: whoever runs first after the idle gap is favoured
    for l in order:
        for cname, k in cases:
            if k:
                st = states[k]
                ms = l.timed(lambda: l.L.vp_jfa_pass(l.ctx, fr, k, st.data_ptr(), None, None, out.data_ptr(), 2), a.reps)
                chk = out
            else:
                st = states[1]
                ms = l.timed(lambda: l.L.vp_jfa_last_pass(l.ctx, fr, st.data_ptr(), None, None, out.data_ptr(), g.data_ptr(), -math.inf, sdf.data_ptr(), 2), a.reps)
                chk = sdf.view(torch.int32)
            if r == 0:                                            # warm-up round: also compare results between the libraries
                l.L.vp_ctx_sync(l.ctx)
                c32 = chk.view(torch.int32)
                h = sum(int(c.to(torch.int64).sum().item()) * (i + 1) for i, c in enumerate(c32.chunk(64))) & (2**63 - 1)
                same = ref.setdefault(cname, h) == h
                print("check %-10s %-8s %s" % (l.name, cname, "same as " + libs[0].name if same else "DIFFERENT RESULT"))
            else:
                res[(l.name, cname)].append(ms)
print("%-14s" % "lib" + "".join("%22s" % c for c, _ in cases) + "      (median / min ms over %d interleaved rounds)" % a.rounds)
for l in libs:
    print("%-14s" % l.name + "".join("%14.4f /%7.4f" % (statistics.median(res[(l.name, c)]), min(res[(l.name, c)])) for c, _ in cases))
