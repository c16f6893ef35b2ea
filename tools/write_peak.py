"""Dev tool: the pure-STORE rate of the box (nothing read): hipMemsetD32 through vp_jfa_window_clear and torch's fill kernel over one id volume --
what a kernel that only writes ids (jfa_first_two: 4 n^3 bytes out, n^3/8 in) can be compared with.   python tools/write_peak.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd.capi import Frame, Window
from cuda_mesh_voxelization_amd.pipeline import Engine
eng = Engine(0)
for n in (512, 1024):
    fr = Frame.make(n, 0.01, (0.0, 0.0, 0.0))
    nb = eng.ctx.jfa_window_bytes(fr, n)
    t = torch.empty(nb, dtype=torch.uint8, device=eng.device)
    w = Window.make(t.data_ptr(), nb, n, 0)
    t32 = t.view(torch.int32)
    for name, fn in (("vp_jfa_window_clear (hipMemsetD32Async)", lambda: eng.ctx.jfa_window_clear(fr, w)), ("torch fill_ (int32)", lambda: t32.fill_(7))):
        for _ in range(3): fn()
        best = 1e9
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("n = %4d  %-40s %8.4f ms  %7.1f GB/s written  (%.3f of 8 TB/s)" % (n, name, best, nb / best / 1e6, nb / best / 1e6 / 8000))
