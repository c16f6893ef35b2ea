"""Dev tool: n = 2048 (64-bit JFA ids) on one GPU -- bitmask vs oracle, tiled JFA vs naive JFA, timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, Frame
from cuda_mesh_voxelization_amd.pipeline import Engine
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
refine = int(sys.argv[2]) if len(sys.argv) > 2 else 24
with_oracle_sdf = len(sys.argv) > 3 and sys.argv[3] == "oracle"
xyz, tri = M.bunny(refine); origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)
print("faces", tri.shape[0], "n", n)
eng = Engine(0); dx, dt = eng.mesh_to_device(xyz, tri)
g = eng.voxelize(fr, dx, dt); eng.sync()
t = time.perf_counter(); g = eng.voxelize(fr, dx, dt); eng.sync(); print("voxelize ms", (time.perf_counter() - t) * 1e3)
t = time.perf_counter(); exp = O.voxelize(xyz, tri, n, vs, origin); print("oracle voxelize s", time.perf_counter() - t)
got = eng.words_to_numpy(g)
print("bitmask equal:", np.array_equal(got, exp), "popcount", O.popcount(got), "fnv", O.fnv(got))
s = eng.jfa(fr, g, algo=ALGO_TILED); eng.sync()
t = time.perf_counter(); s = eng.jfa(fr, g, algo=ALGO_TILED); eng.sync(); print("jfa tiled ms", (time.perf_counter() - t) * 1e3)
chk_t = int(s.view(torch.int32).to(torch.int64).sum().item())
s_t = s.clone()
t = time.perf_counter(); s_n = eng.jfa(fr, g, algo=ALGO_NAIVE); eng.sync(); print("jfa naive ms", (time.perf_counter() - t) * 1e3)
print("tiled == naive:", bool(torch.equal(s_t.view(torch.int32), s_n.view(torch.int32))), "checksum", chk_t)
del s_n
zeros = sum(int((s_t[i:i + (1 << 30)] == 0).sum().item()) for i in range(0, s_t.numel(), 1 << 30))
print("zeros", zeros)
if with_oracle_sdf:
    t = time.perf_counter(); e = O.jfa(exp, n, vs, origin); print("oracle jfa s", time.perf_counter() - t, "threads", O.threads())
    h = s_t.cpu().numpy()
    print("sdf == oracle:", np.array_equal(h.view(np.uint32), e.view(np.uint32)), "fnv", O.fnv(h), O.fnv(e))
