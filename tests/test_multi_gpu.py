"""vp_multi_* -- the C++ Z-slab driver behind the C ABI (csrc/multi.hip; what replaces the reference's hard-wired device 0,
apps/cli/main.cpp:22-23).  A one-GPU box runs it with several contexts on device 0: the slab frames, halo copies, stream events
and ghost regions are the ones a multi-device node runs, only the copies stay on the device.  Bar: concatenated slabs
bit-identical to the single-context result (and, at the small sizes, to the oracle)."""
import numpy as np
import pytest
import torch

from cuda_mesh_voxelization_amd import capi, mesh as M
from cuda_mesh_voxelization_amd.capi import ALGO_NAIVE, ALGO_TILED, MULTI_GHOST, MULTI_HALO, MULTI_HYBRID, MULTI_TRANSPOSE, Frame
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _single(engine, fr, xyz, tri, algo=ALGO_TILED):
    dx, dt = engine.mesh_to_device(xyz, tri)
    g = engine.voxelize(fr, dx, dt, algo=algo)
    s = engine.jfa(fr, g, algo=algo)
    engine.sync()
    return engine.words_to_numpy(g).copy(), s.cpu().numpy()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name,n", [("bunny.obj", 64), ("bunny.obj", 256)])
def test_multi_matches_single_and_oracle(engine, name, n, world):
    xyz, tri = M.import_mesh(M.asset(name))
    origin, vs = M.frame([xyz], n)
    fr = Frame.make(n, vs, origin)
    exp_w = O.voxelize(xyz, tri, n, vs, origin)
    exp_s = O.jfa(exp_w, n, vs, origin)
    m = capi.Multi([0] * world)
    try:
        assert m.count == world
        m.set_mesh(xyz, tri)
        for algo in (ALGO_TILED, ALGO_NAIVE):
            m.voxelize(fr, algo)
            assert np.array_equal(m.get_grid(), exp_w)
            vp_id_bytes = 4                                             # n <= 1024
            for mode in (MULTI_HALO, MULTI_GHOST, MULTI_HYBRID, MULTI_TRANSPOSE):
                m.jfa(algo=algo, mode=mode)
                got = m.get_sdf()
                assert np.array_equal(got.view(np.uint32), exp_s.view(np.uint32)), (algo, mode)
                # halo: planes really moved (bitmask planes + id planes of every pass); ghost: only the bitmask all-gather;
                # hybrid: the all-gather + k planes per side and narrow pass
                gather = world * (world - 1) * (fr.words // world) * 4
                if n < 96:                                               # below the tile kernels' range the grid is not sharded: every device
                    assert m.bytes_moved == gather and m.window(0)[:2] == (0, n)    # computes the whole 64^3 grid and keeps its slab
                elif mode == MULTI_GHOST:
                    assert m.bytes_moved == gather
                elif mode == MULTI_HYBRID:
                    nz = n // world
                    narrow = [k for k in (n >> i for i in range(1, n.bit_length())) if 1 <= k <= nz // 2]
                    assert m.bytes_moved == gather + 2 * (world - 1) * sum(narrow) * n * n * 4
                    for r in range(world):                               # the id volumes hold the rank's window, not the grid
                        lo, hi, nbytes = m.window(r)
                        assert 0 <= lo <= r * nz and (r + 1) * nz <= hi <= n and nbytes >= 2 * (hi - lo) * n * n * vp_id_bytes
                elif mode == MULTI_TRANSPOSE:
                    # the all-gather + ONE re-deal: every device receives the planes of its widened slab the others hold
                    from cuda_mesh_voxelization_amd.slab import transpose_plan
                    plans = [transpose_plan(n, r, world) for r in range(world)]
                    assert m.bytes_moved == gather + sum((world - 1) * ((p["recv"][1] - p["recv"][0]) // world) * n * n * 4 for p in plans)
                    for r, p in enumerate(plans):
                        lo, hi, nbytes = m.window(r)
                        assert (lo, hi) == p["window"]
                        assert nbytes == (2 * (hi - lo) + 2 * (n // world) + (p["recv"][1] - p["recv"][0])) * n * n * vp_id_bytes
                else:
                    assert m.bytes_moved > 2 * (world - 1) * n * n * 4
    finally:
        m.close()


def test_multi_csg_and_set_grid(engine):
    """config 3's flow on slabs: two meshes in one frame, union accumulated into the resident grid, then the sdf."""
    a = M.import_mesh(M.asset("bimba.obj"))
    b = M.import_mesh(M.asset("bunny.obj"))
    n, world = 128, 4
    origin, vs = M.frame([a[0], b[0]], n)
    fr = Frame.make(n, vs, origin)
    wa, wb = O.voxelize(a[0], a[1], n, vs, origin), O.voxelize(b[0], b[1], n, vs, origin)
    m = capi.Multi([0] * world)
    try:
        for op in (1, 2, 3):
            m.set_mesh(*a)
            m.voxelize(fr)
            m.csg(wb, op)
            exp = wa.copy()
            O.csg(exp, wb, op)
            assert np.array_equal(m.get_grid(), exp), op
            m.jfa(mode=MULTI_HALO)
            assert np.array_equal(m.get_sdf().view(np.uint32), O.jfa(exp, n, vs, origin).view(np.uint32)), op
        # a host grid scattered into the slabs gives the same sdf as the one voxelized in place
        m.set_grid(fr, exp)
        m.jfa(mode=MULTI_GHOST)
        assert np.array_equal(m.get_sdf().view(np.uint32), O.jfa(exp, n, vs, origin).view(np.uint32))
    finally:
        m.close()


@pytest.mark.parametrize("world,mode", [(2, MULTI_HALO), (4, MULTI_HALO), (8, MULTI_HALO), (4, MULTI_GHOST), (8, MULTI_GHOST), (4, MULTI_HYBRID), (8, MULTI_HYBRID),
                                        (2, MULTI_TRANSPOSE), (4, MULTI_TRANSPOSE), (8, MULTI_TRANSPOSE)])
def test_multi_headline_size_equals_single(engine, world, mode):
    """n = 512 on the benchmark mesh: halos of the narrow passes land next to the slab, the slabs of the wide ones a slab height away
    (stride = nz); ghost regions take the first two passes as the one whole-grid launch."""
    xyz, tri = M.bunny(24)
    n = 512
    origin, vs = M.frame([xyz], n)
    fr = Frame.make(n, vs, origin)
    ref_w, ref_s = _single(engine, fr, xyz, tri)
    m = capi.Multi([0] * world)
    try:
        m.set_mesh(xyz, tri)
        m.voxelize(fr)
        assert np.array_equal(m.get_grid(), ref_w)
        m.jfa(mode=mode)
        assert np.array_equal(m.get_sdf().view(np.uint32), ref_s.view(np.uint32))
    finally:
        m.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", [MULTI_HALO, MULTI_GHOST])
def test_multi_config4_n1024_four_slabs(engine, mode):
    """BASELINE config 4 at its stated shape through the C++ driver: 1,348,128 faces, n = 1024, four Z-slabs (four contexts on the
    one device of the test box): grid and sdf bit-identical to the single-context result."""
    import gc
    xyz, tri = M.bunny(24)
    n, world = 1024, 4
    origin, vs = M.frame([xyz], n)
    fr = Frame.make(n, vs, origin)
    ref_w, ref_s = _single(engine, fr, xyz, tri)
    engine._work = None
    gc.collect(); torch.cuda.empty_cache()
    m = capi.Multi([0] * world)
    try:
        m.set_mesh(xyz, tri)
        m.voxelize(fr)
        assert np.array_equal(m.get_grid(), ref_w)
        m.jfa(mode=mode)
        assert np.array_equal(m.get_sdf().view(np.uint32), ref_s.view(np.uint32))
        if mode == MULTI_HALO:                                      # ~5 GiB received per interior rank over the 10 passes
            assert 3 * 2**30 < m.bytes_moved / world < 6 * 2**30
    finally:
        m.close()
        gc.collect(); torch.cuda.empty_cache()


@pytest.mark.parametrize("poison", ["0xA5", "0xFF"])
def test_multi_ghost_ignores_unproduced_planes(engine, poison):
    """the ghost regions are rounded outwards to the 8-plane tile, so the excess planes of a pass read planes the pass
    before it never produced.  In the hooks build (libvphip_hooks.so, VP_MULTI_POISON) the word planes of the id windows are refilled
    with a poison byte before every vp_multi_jfa: the slabs must not depend on it.  n = 256 over 8 ranks: slabs of 32 planes, every pass
    with k < 8 has rounding excess on both sides; n = 1152 over 4: the 5-byte window layout, the windows re-used by another mode (another
    geometry: cleared) in between.  VP_MULTI_TRANSPOSE: the planes around the received range of the slab windows are never written
    (the rounded regions of the last passes read them), its cyclic and staging windows are poisoned too."""
    import os, subprocess, sys
    from cuda_mesh_voxelization_amd import build
    build.build_lib(hooks=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from cuda_mesh_voxelization_amd import capi, mesh as M\n"
        "from cuda_mesh_voxelization_amd.capi import Frame, MULTI_GHOST, MULTI_HALO, MULTI_HYBRID, MULTI_TRANSPOSE\n"
        "from cuda_mesh_voxelization_amd.pipeline import Engine\n"
        "eng = Engine(0)\n"
        "xyz, tri = M.import_mesh(M.asset('bunny.obj'))\n"
        "for n, world in ((256, 8), (1152, 4)):\n"
        "    origin, vs = M.frame([xyz], n); fr = Frame.make(n, vs, origin)\n"
        "    dx, dt = eng.mesh_to_device(xyz, tri)\n"
        "    ref = eng.jfa(fr, eng.voxelize(fr, dx, dt)).cpu().numpy(); eng._work = None\n"
        "    m = capi.Multi([0] * world)\n"
        "    m.set_mesh(xyz, tri); m.voxelize(fr)\n"
        "    for mode in (MULTI_GHOST, MULTI_TRANSPOSE, MULTI_HYBRID, MULTI_GHOST, MULTI_HALO, MULTI_TRANSPOSE):\n"
        "        m.jfa(mode=mode)\n"
        "        assert np.array_equal(m.get_sdf().view(np.uint32), ref.view(np.uint32)), (n, mode)\n"
        "    m.close()\n"
        "print('ok')\n" % root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, VPHIP_LIB=capi.HOOKS_LIB_PATH, VP_MULTI_POISON=poison))
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout[-500:], p.stderr[-3000:])


def test_python_ghost_and_hybrid_ignore_unproduced_planes(engine):
    """the same for the one-process-per-GPU pipelines of slab.py (HipSlabBackend(poison=...) overwrites the word planes of fresh windows)"""
    from cuda_mesh_voxelization_amd.slab import GhostSlabPipeline, HipSlabBackend, HybridSlabPipeline

    class _Alone:                                                    # the hybrid pipeline of rank 0 of 1 exchanges nothing
        pass

    xyz, tri = M.import_mesh(M.asset("bunny.obj"))
    n, world = 256, 8
    origin, vs = M.frame([xyz], n)
    fr = Frame.make(n, vs, origin)
    _, ref_s = _single(engine, fr, xyz, tri)
    dx, dt = engine.mesh_to_device(xyz, tri)
    nzv = fr.voxels // world
    for r in range(world):
        pipe = GhostSlabPipeline(HipSlabBackend(engine, poison=0xA5), fr, r, world)
        pipe.voxelize(dx, dt)
        s = pipe.jfa().cpu().numpy()
        assert np.array_equal(s.view(np.uint32), ref_s[r * nzv:(r + 1) * nzv].view(np.uint32)), r
        assert pipe.report()["hbm_bytes_this_rank"] >= 2 * fr.voxels * 4 + fr.words * 4
    pipe = HybridSlabPipeline(HipSlabBackend(engine, poison=0xA5), fr, 0, 1, _Alone())
    pipe.voxelize(dx, dt)
    assert np.array_equal(pipe.jfa().cpu().numpy().view(np.uint32), ref_s.view(np.uint32))


def test_multi_csg_checks_the_operand_size(engine):
    """vp_multi_csg takes the word count of its host operand, like vp_csg (csg/naive.cu:30-33: equal grids)."""
    fr = Frame.make(64, 1.0, (0, 0, 0))
    m = capi.Multi([0, 0])
    try:
        m.set_grid(fr, np.zeros(fr.words, np.uint32))
        with pytest.raises(capi.VPError, match="words given"):
            m.csg(np.zeros(fr.words // 2, np.uint32), 1)
        m.csg(np.ones(fr.words, np.uint32), 1)
        assert np.array_equal(m.get_grid(), np.ones(fr.words, np.uint32))
        # a failing replacement of the grid must not leave the old one flagged as resident under the new frame
        with pytest.raises(capi.VPError):
            m.voxelize(Frame.make(100, 1.0, (0, 0, 0)))            # n % 32 != 0: refused by check_split, BEFORE anything is touched
        assert np.array_equal(m.get_grid(), np.ones(fr.words, np.uint32))
    finally:
        m.close()


def test_multi_hybrid_n1024_eight_slabs_window(engine):
    """windowed id state in the C++ slab driver.  VP_MULTI_HYBRID on eight contexts of the one device, the benchmark
    mesh at n = 1024 (eight windows of a real node sit on eight devices; on ONE device they have to fit together, which at n = 2048
    -- 8 x 112 GiB -- they cannot: that size is covered rank by rank with the Python pipelines, tests/test_slab_gpu.py): grid and sdf
    bit-identical to the single-context result, every rank's id volumes cover its window only."""
    import gc
    gc.collect(); torch.cuda.empty_cache()
    xyz, tri = M.bunny(24)
    n, world = 1024, 8
    origin, vs = M.frame([xyz], n)
    fr = Frame.make(n, vs, origin)
    ref_w, ref_s = _single(engine, fr, xyz, tri)
    engine._work = None
    gc.collect(); torch.cuda.empty_cache()
    nz = n // world
    m = capi.Multi([0] * world)
    try:
        m.set_mesh(xyz, tri)
        m.voxelize(fr)
        assert np.array_equal(m.get_grid(), ref_w)
        m.jfa(mode=MULTI_HYBRID)
        assert np.array_equal(m.get_sdf().view(np.uint32), ref_s.view(np.uint32))
        total = 0
        for r in range(world):
            lo, hi, nbytes = m.window(r)
            assert lo <= r * nz and (r + 1) * nz <= hi and hi - lo <= 896, (r, lo, hi)       # 896 of 1024 planes at most
            assert nbytes == 2 * (hi - lo) * n * n * 4                                        # exactly the window, twice
            total += nbytes
        assert total <= 0.875 * world * 2 * n * n * n * 4                                     # against eight pairs of whole volumes
        m.jfa(mode=MULTI_GHOST)                                                               # the buffers grow to whole volumes and the answer stays
        assert np.array_equal(m.get_sdf().view(np.uint32), ref_s.view(np.uint32))
        assert m.window(3)[:2] == (0, n)
    finally:
        m.close()
        gc.collect(); torch.cuda.empty_cache()


def test_multi_rejects_bad_splits_and_order(engine):
    fr = Frame.make(96, 1.0, (0, 0, 0))
    m = capi.Multi([0, 0, 0, 0, 0])
    try:
        with pytest.raises(capi.VPError, match="cannot be cut"):
            m.set_grid(fr, np.zeros(fr.words, np.uint32))
        m.frame = fr
        with pytest.raises(capi.VPError, match="no resident grid"):
            m.jfa()
        with pytest.raises(capi.VPError, match="no sdf"):
            m.get_sdf()
    finally:
        m.close()
    with pytest.raises(capi.VPError, match="not present"):
        capi.Multi([0, 99])


@pytest.mark.parametrize("world,n,name", [(8, 1024, None), (4, 1152, "bimba.obj"), (3, 192, "torus.obj")])
def test_multi_transpose_equals_single(engine, world, n, name):
    """VP_MULTI_TRANSPOSE in the one-process driver (peer copies of contiguous plane ranges straight from the windows of the cyclic phase):
    the benchmark mesh at n = 1024 on eight contexts of the one device -- 0.49 GiB into each device where the halo mode moves 3.5 GiB --,
    the compact windows with a step sequence that leaves the multiples of four early (n = 1152), and three devices (not a power of two:
    the mode runs ghost planes and reports it)."""
    import gc
    gc.collect(); torch.cuda.empty_cache()
    xyz, tri = M.bunny(24) if name is None else M.import_mesh(M.asset(name))
    origin, vs = M.frame([xyz], n)
    fr = Frame.make(n, vs, origin)
    ref_w, ref_s = _single(engine, fr, xyz, tri)
    engine._work = None
    gc.collect(); torch.cuda.empty_cache()
    m = capi.Multi([0] * world)
    try:
        m.set_mesh(xyz, tri)
        m.voxelize(fr)
        m.jfa(mode=MULTI_TRANSPOSE)
        assert np.array_equal(m.get_sdf().view(np.uint32), ref_s.view(np.uint32))
        assert np.array_equal(m.get_grid(), ref_w)
        gather = world * (world - 1) * (fr.words // world) * 4
        if world == 3:
            assert m.bytes_moved == gather and m.window(0)[:2] == (0, n)            # ghost planes: whole volumes, nothing but the all-gather
        else:
            S = 5 if n > 1024 else 4
            per_rank = [(m.bytes_moved - gather) / world]
            from cuda_mesh_voxelization_amd.slab import transpose_plan
            want = sum((world - 1) * ((p["recv"][1] - p["recv"][0]) // world) * n * n * S for p in (transpose_plan(n, r, world) for r in range(world)))
            assert m.bytes_moved == gather + want
            if n == 1024:
                assert max(per_rank) <= 0.6 * 2**30
        m.jfa(mode=MULTI_GHOST)                                                       # another mode on the same driver afterwards
        assert np.array_equal(m.get_sdf().view(np.uint32), ref_s.view(np.uint32))
    finally:
        m.close()
        gc.collect(); torch.cuda.empty_cache()
