"""Dev tool: where a tile of jfa_first_two spends its time.  Needs a library built with -DVP_FIRST_TWO_TIMING
(tools/exp_build.sh fttime -DVP_FIRST_TWO_TIMING): thread 0 of every workgroup stamps s_memtime at the stage boundaries.
  VPHIP_LIB=tools/exp/libvphip_fttime.so python tools/first_two_stages.py [n]"""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_mesh_voxelization_amd import capi, mesh as M
from cuda_mesh_voxelization_amd.capi import Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
xyz, tri = M.bunny(24); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.voxelize(fr, dx, dt)
border = eng.surface(fr, g)
ids = torch.empty(fr.voxels * (eng.ctx.jfa_id_bytes(fr) // 4), dtype=torch.int32, device=eng.device)
L = capi.lib()
L.vp_dev_first_two_times.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
acc = (ctypes.c_uint64 * 16)()
for _ in range(3): eng.ctx.jfa_first_two(fr, border.data_ptr(), ids.data_ptr())
eng.sync(); L.vp_dev_first_two_times(acc, 1)
reps = 1
eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
for _ in range(reps): eng.ctx.jfa_first_two(fr, border.data_ptr(), ids.data_ptr())
eng.ctx.prof_enable(False); eng.sync()
ms = eng.ctx.prof()["jfa_first"]; L.vp_dev_first_two_times(acc, 0)
names = ["", "init cnt + barrier", "border words -> flags, keys/idOf init", "append A (ballot, atomic, list)", "barrier", "scatter A (k = n/2)", "barrier",
         "collect A (keys -> seedOf), reset", "append B", "barrier", "scatter B (k = n/4)", "barrier", "output (keys -> ids -> store issue)"]
wgs = acc[0]
print("n = %d: %d workgroups in %d launches, kernel %.3f ms per launch; s_memtime ticks taken as 100 MHz (10 ns); one row per workgroup (the last launch), so sums are over min(workgroups, 131072) workgroups" % (n, wgs, reps, ms["ms"] / ms["launches"]))
tot = sum(acc[i] for i in range(1, 13))
for i in range(1, 13):
    print("  %-48s %8.1f ns  %5.1f %%" % (names[i], acc[i] / wgs * 10.0, 100.0 * acc[i] / tot))
print("  %-48s %8.1f ns per workgroup (thread 0, first stamp to last)" % ("total", tot / wgs * 10.0))
