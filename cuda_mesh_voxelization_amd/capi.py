"""ctypes binding of libvphip.so (include/vphip.h) -- the only way the Python harness reaches
the HIP kernels.  There is no fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# VPHIP_LIB: another build of the library -- dev experiments (tools/exp_build.sh) and the tests that need libvphip_hooks.so
LIB_PATH = os.environ.get("VPHIP_LIB") or os.path.join(PKG, "libvphip.so")
HOOKS_LIB_PATH = os.path.join(PKG, "libvphip_hooks.so")

ALGO_NAIVE, ALGO_TILED = 1, 2
OP_VOID, OP_UNION, OP_INTERSECTION, OP_DIFFERENCE = 0, 1, 2, 3
EXTRACT_SET, EXTRACT_EXPOSED, EXTRACT_FACES = 0, 1, 2
MULTI_HALO, MULTI_GHOST, MULTI_HYBRID, MULTI_TRANSPOSE = 0, 1, 2, 3

KERNELS = ["vox_setup", "vox_scan", "vox_scatter", "vox_tile", "vox_naive", "vox_fill",
           "csg_words", "jfa_init", "jfa_pass", "jfa_final", "surface", "jfa_first", "jfa_sparse", "jfa_dense", "jfa_last", "extract", "vox_zero", "jfa_redeal"]
JFA_PASS_KEYS = ("jfa_pass", "jfa_first", "jfa_sparse", "jfa_dense", "jfa_last")

# every symbol include/vphip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "vp_device_count", "vp_ctx_create", "vp_ctx_destroy", "vp_ctx_set_stream", "vp_ctx_sync", "vp_last_error", "vp_abi_version",
    "vp_malloc", "vp_free", "vp_memset", "vp_memcpy_d2d", "vp_stream_copy", "vp_ctx_workspace", "vp_ctx_release", "vp_upload", "vp_download",
    "vp_grid_words", "vp_grid_voxels",
    "vp_voxelize", "vp_csg", "vp_jfa_workspace_bytes", "vp_jfa_id_bytes", "vp_jfa_state_bytes", "vp_jfa", "vp_jfa_start", "vp_jfa_run", "vp_jfa_init", "vp_jfa_pass",
    "vp_jfa_finalize", "vp_jfa_last_pass", "vp_jfa_can_start_from_mask", "vp_jfa_can_fuse_first_two",
    "vp_jfa_window_bytes", "vp_jfa_window_span", "vp_jfa_window_clear", "vp_jfa_window_init", "vp_jfa_window_first_pass", "vp_jfa_window_first_two",
    "vp_jfa_window_pass", "vp_jfa_window_last_pass",
    "vp_jfa_cyclic_passes", "vp_jfa_window_first_two_cyclic", "vp_jfa_window_pass_cyclic", "vp_jfa_window_interleave",
    "vp_surface", "vp_extract_count", "vp_extract", "vp_voxelize_host", "vp_csg_host", "vp_jfa_host",
    "vp_prof_enable", "vp_prof_select", "vp_prof_reset", "vp_prof_get", "vp_prof_name",
    "vp_multi_create", "vp_multi_destroy", "vp_multi_count", "vp_multi_ctx", "vp_multi_sync", "vp_multi_set_mesh", "vp_multi_voxelize",
    "vp_multi_set_grid", "vp_multi_get_grid", "vp_multi_csg", "vp_multi_jfa", "vp_multi_get_sdf", "vp_multi_bytes_moved", "vp_multi_window",
]


class VPError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libvphip error %d: %s" % (code, msg))
        self.code = code


class Frame(ctypes.Structure):
    """vp_frame: global grid side n, voxel size, origin, Z-slab [z0, z1)."""
    _fields_ = [("n", ctypes.c_uint32), ("voxel_size", ctypes.c_float), ("origin", ctypes.c_float * 3),
                ("z0", ctypes.c_uint32), ("z1", ctypes.c_uint32)]

    @classmethod
    def make(cls, n, voxel_size, origin, z0=0, z1=None):
        f = cls()
        f.n = int(n)
        f.voxel_size = float(voxel_size)
        f.origin[0], f.origin[1], f.origin[2] = float(origin[0]), float(origin[1]), float(origin[2])
        f.z0 = int(z0)
        f.z1 = int(n if z1 is None else z1)
        return f

    @property
    def nz(self):
        return self.z1 - self.z0

    @property
    def words(self):
        return self.n * self.n * self.nz // 32

    @property
    def voxels(self):
        return self.n * self.n * self.nz

    def slab(self, z0, z1):
        return Frame.make(self.n, self.voxel_size, self.origin, z0, z1)


class Window(ctypes.Structure):
    """vp_window: an id buffer in the library's layout -- `planes` id planes, plane z0 of the frame a call is made with at index `at`."""
    _fields_ = [("d_ids", ctypes.c_void_p), ("bytes", ctypes.c_size_t), ("planes", ctypes.c_uint32), ("at", ctypes.c_uint32)]

    @classmethod
    def make(cls, d_ids, nbytes, planes, at):
        """nbytes: the size of the buffer (every call checks it against vp_jfa_window_bytes of `planes` planes)"""
        w = cls()
        w.d_ids, w.bytes, w.planes, w.at = int(d_ids), int(nbytes), int(planes), int(at)
        return w


_lib = None
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t


def lib():
    """Load libvphip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libvphip.so is missing at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
    # Load order: torch ships its own libamdhip64.so, libvphip.so is linked against the one of /opt/rocm.  Whichever is mapped first serves
    # both (same SONAME); if libvphip.so comes first and torch initialises the GPU afterwards, the process ends up with two HIP runtimes
    # and vp_ctx_create finds no device (seen on the GPU box with `python __graft_entry__.py smoke`, which builds -- and loads -- first).
    # The harness always runs beside torch, so torch goes first here.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    fp = ctypes.POINTER(Frame)
    wp = ctypes.POINTER(Window)
    sig = {
        "vp_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
        "vp_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
        "vp_ctx_destroy": (ctypes.c_int, [_vp]),
        "vp_ctx_set_stream": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
        "vp_ctx_sync": (ctypes.c_int, [_vp]),
        "vp_last_error": (ctypes.c_char_p, []),
        "vp_abi_version": (ctypes.c_int, []),
        "vp_malloc": (ctypes.c_int, [_vp, _sz, ctypes.POINTER(_vp)]),
        "vp_free": (ctypes.c_int, [_vp, _vp]),
        "vp_memset": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _sz]),
        "vp_memcpy_d2d": (ctypes.c_int, [_vp, _vp, _vp, _sz]),
        "vp_stream_copy": (ctypes.c_int, [_vp, _vp, _vp, _sz]),
        "vp_ctx_workspace": (ctypes.c_int, [_vp, ctypes.c_int, _sz, ctypes.POINTER(_vp)]),
        "vp_ctx_release": (ctypes.c_int, [_vp]),
        "vp_upload": (ctypes.c_int, [_vp, _vp, _vp, _sz]),
        "vp_download": (ctypes.c_int, [_vp, _vp, _vp, _sz]),
        "vp_grid_words": (_sz, [fp]),
        "vp_grid_voxels": (_sz, [fp]),
        "vp_voxelize": (ctypes.c_int, [_vp, fp, _vp, _vp, _sz, _vp, _sz, ctypes.c_int, ctypes.c_int]),
        "vp_csg": (ctypes.c_int, [_vp, _vp, _vp, _sz, ctypes.c_int]),
        "vp_jfa_workspace_bytes": (_sz, [fp]),
        "vp_jfa_id_bytes": (_sz, [fp]),
        "vp_jfa_state_bytes": (_sz, [fp, ctypes.c_int]),
        "vp_jfa": (ctypes.c_int, [_vp, fp, _vp, ctypes.c_float, _vp, _vp, _sz, ctypes.c_int]),
        "vp_jfa_start": (ctypes.c_int, [_vp, fp, _vp, _vp, _sz, ctypes.c_int]),
        "vp_jfa_run": (ctypes.c_int, [_vp, fp, _vp, ctypes.c_float, _vp, _vp, _sz, ctypes.c_int]),
        "vp_jfa_init": (ctypes.c_int, [_vp, fp, _vp, _vp, _vp, _vp]),
        "vp_jfa_pass": (ctypes.c_int, [_vp, fp, ctypes.c_uint32, _vp, _vp, _vp, _vp, ctypes.c_int]),
        "vp_jfa_finalize": (ctypes.c_int, [_vp, fp, _vp, _vp, ctypes.c_float, _vp]),
        "vp_jfa_last_pass": (ctypes.c_int, [_vp, fp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int]),
        "vp_jfa_can_start_from_mask": (ctypes.c_int, [fp, ctypes.c_int]),
        "vp_jfa_can_fuse_first_two": (ctypes.c_int, [fp, ctypes.c_int]),
        "vp_jfa_window_bytes": (_sz, [fp, ctypes.c_uint32]),
        "vp_jfa_window_span": (ctypes.c_int, [fp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
        "vp_jfa_window_clear": (ctypes.c_int, [_vp, fp, wp]),
        "vp_jfa_window_init": (ctypes.c_int, [_vp, fp, _vp, _vp, _vp, wp]),
        "vp_jfa_window_first_pass": (ctypes.c_int, [_vp, fp, _vp, wp]),
        "vp_jfa_window_first_two": (ctypes.c_int, [_vp, fp, _vp, wp]),
        "vp_jfa_window_pass": (ctypes.c_int, [_vp, fp, ctypes.c_uint32, wp, wp, ctypes.c_uint32]),
        "vp_jfa_window_last_pass": (ctypes.c_int, [_vp, fp, wp, wp, ctypes.c_uint32, _vp, ctypes.c_float, _vp]),
        "vp_jfa_cyclic_passes": (ctypes.c_int, [fp, ctypes.c_uint32]),
        "vp_jfa_window_first_two_cyclic": (ctypes.c_int, [_vp, fp, _vp, wp, ctypes.c_uint32, ctypes.c_uint32]),
        "vp_jfa_window_pass_cyclic": (ctypes.c_int, [_vp, fp, ctypes.c_uint32, wp, wp, ctypes.c_uint32, ctypes.c_uint32]),
        "vp_jfa_window_interleave": (ctypes.c_int, [_vp, fp, wp, wp, ctypes.c_uint32, ctypes.c_uint32]),
        "vp_surface": (ctypes.c_int, [_vp, fp, _vp, _vp, _vp, _vp]),
        "vp_extract_count": (ctypes.c_int, [_vp, fp, _vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
        "vp_extract": (ctypes.c_int, [_vp, fp, _vp, ctypes.c_int, _vp, _vp, _vp, _sz]),
        "vp_voxelize_host": (ctypes.c_int, [_vp, fp, _vp, _vp, _sz, _vp, _sz, ctypes.c_int]),
        "vp_csg_host": (ctypes.c_int, [_vp, _vp, _vp, _sz, ctypes.c_int]),
        "vp_jfa_host": (ctypes.c_int, [_vp, fp, _vp, ctypes.c_float, _vp, ctypes.c_int]),
        "vp_prof_enable": (ctypes.c_int, [_vp, ctypes.c_int]),
        "vp_prof_select": (ctypes.c_int, [_vp, ctypes.c_uint64]),
        "vp_prof_reset": (ctypes.c_int, [_vp]),
        "vp_prof_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]),
        "vp_prof_name": (ctypes.c_char_p, [ctypes.c_int]),
        "vp_multi_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(_vp)]),
        "vp_multi_destroy": (ctypes.c_int, [_vp]),
        "vp_multi_count": (ctypes.c_int, [_vp]),
        "vp_multi_ctx": (_vp, [_vp, ctypes.c_int]),
        "vp_multi_sync": (ctypes.c_int, [_vp]),
        "vp_multi_set_mesh": (ctypes.c_int, [_vp, _vp, _sz, _vp, _sz]),
        "vp_multi_voxelize": (ctypes.c_int, [_vp, fp, ctypes.c_int]),
        "vp_multi_set_grid": (ctypes.c_int, [_vp, fp, _vp]),
        "vp_multi_get_grid": (ctypes.c_int, [_vp, _vp]),
        "vp_multi_csg": (ctypes.c_int, [_vp, _vp, _sz, ctypes.c_int]),
        "vp_multi_jfa": (ctypes.c_int, [_vp, ctypes.c_float, ctypes.c_int, ctypes.c_int]),
        "vp_multi_get_sdf": (ctypes.c_int, [_vp, _vp]),
        "vp_multi_bytes_moved": (ctypes.c_uint64, [_vp]),
        "vp_multi_window": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise VPError(rc, (lib().vp_last_error() or b"").decode("utf-8", "replace"))


class Context:
    """vp_ctx wrapper.  Device pointers are plain ints (e.g. torch.Tensor.data_ptr())."""

    def __init__(self, device: int = 0):
        self._h = _vp()
        check(lib().vp_ctx_create(device, ctypes.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().vp_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing
    def set_stream(self, hip_stream, external: bool = True):
        """external=True: run on the caller's stream handle (0 = the null stream = torch's default stream)."""
        check(lib().vp_ctx_set_stream(self._h, _vp(hip_stream) if hip_stream else None, 1 if external else 0))

    def sync(self):
        check(lib().vp_ctx_sync(self._h))

    def malloc(self, nbytes: int) -> int:
        p = _vp()
        check(lib().vp_malloc(self._h, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, ptr: int):
        check(lib().vp_free(self._h, _vp(ptr)))

    def memset(self, ptr: int, value: int, nbytes: int):
        check(lib().vp_memset(self._h, _vp(ptr), value, nbytes))

    def upload(self, dptr: int, host_array):
        check(lib().vp_upload(self._h, _vp(dptr), host_array.ctypes.data_as(_vp), host_array.nbytes))

    def download(self, host_array, dptr: int):
        check(lib().vp_download(self._h, host_array.ctypes.data_as(_vp), _vp(dptr), host_array.nbytes))

    # -- device-resident stages
    def voxelize(self, frame: Frame, d_words: int, d_xyz: int, nverts: int, d_tri: int, ntris: int,
                 algo: int = ALGO_TILED, accumulate: bool = False):
        check(lib().vp_voxelize(self._h, ctypes.byref(frame), _vp(d_words), _vp(d_xyz), nverts, _vp(d_tri), ntris,
                                algo, 1 if accumulate else 0))

    def csg(self, d_a: int, d_b: int, nwords: int, op: int):
        check(lib().vp_csg(self._h, _vp(d_a), _vp(d_b), nwords, op))

    def jfa_workspace_bytes(self, frame: Frame) -> int:
        return int(lib().vp_jfa_workspace_bytes(ctypes.byref(frame)))

    def jfa_id_bytes(self, frame: Frame) -> int:
        """Bytes of a PLAIN id (the caller-addressed planes of vp_jfa_init / vp_jfa_pass): 4 for n <= 1024, 8 for n <= 2048."""
        return int(lib().vp_jfa_id_bytes(ctypes.byref(frame)))

    def jfa_state_bytes(self, frame: Frame, algo: int = ALGO_TILED) -> int:
        """Bytes of id state per voxel vp_jfa streams per pass (whole grid): 4, 5 (windows above n = 1024) or 8 (VP_ALGO_NAIVE there)."""
        return int(lib().vp_jfa_state_bytes(ctypes.byref(frame), algo))

    def jfa(self, frame: Frame, d_words: int, fill: float, d_sdf: int, d_work=None, work_bytes: int = 0,
            algo: int = ALGO_TILED):
        """d_work = None: the context's own grow-only workspace."""
        check(lib().vp_jfa(self._h, ctypes.byref(frame), _vp(d_words), fill, _vp(d_sdf), _vp(d_work or None), work_bytes, algo))

    def jfa_start(self, frame: Frame, d_words: int, d_work=None, work_bytes: int = 0, algo: int = ALGO_TILED):
        check(lib().vp_jfa_start(self._h, ctypes.byref(frame), _vp(d_words), _vp(d_work or None), work_bytes, algo))

    def jfa_run(self, frame: Frame, d_words: int, fill: float, d_sdf: int, d_work=None, work_bytes: int = 0, algo: int = ALGO_TILED):
        check(lib().vp_jfa_run(self._h, ctypes.byref(frame), _vp(d_words), fill, _vp(d_sdf), _vp(d_work or None), work_bytes, algo))

    def memcpy_d2d(self, dst: int, src: int, nbytes: int):
        check(lib().vp_memcpy_d2d(self._h, _vp(dst), _vp(src), nbytes))

    def stream_copy(self, dst: int, src: int, nbytes: int):
        """the 16-bytes-per-lane copy kernel bench.py measures the box's HBM copy rate with"""
        check(lib().vp_stream_copy(self._h, _vp(dst), _vp(src), nbytes))

    def workspace(self, slot: int, nbytes: int) -> int:
        p = _vp()
        check(lib().vp_ctx_workspace(self._h, slot, nbytes, ctypes.byref(p)))
        return p.value

    def release(self):
        check(lib().vp_ctx_release(self._h))

    def jfa_init(self, frame: Frame, d_words: int, d_below, d_above, d_ids: int):
        check(lib().vp_jfa_init(self._h, ctypes.byref(frame), _vp(d_words), _vp(d_below or None),
                                _vp(d_above or None), _vp(d_ids)))

    def jfa_pass(self, frame: Frame, k: int, d_in: int, d_minus, d_plus, d_out: int, algo: int = ALGO_TILED):
        check(lib().vp_jfa_pass(self._h, ctypes.byref(frame), k, _vp(d_in), _vp(d_minus or None),
                                _vp(d_plus or None), _vp(d_out), algo))

    def jfa_finalize(self, frame: Frame, d_words: int, d_ids: int, fill: float, d_sdf: int):
        check(lib().vp_jfa_finalize(self._h, ctypes.byref(frame), _vp(d_words), _vp(d_ids), fill, _vp(d_sdf)))

    def jfa_last_pass(self, frame: Frame, d_in: int, d_minus, d_plus, d_scratch: int, d_words: int, fill: float, d_sdf: int,
                      algo: int = ALGO_TILED):
        check(lib().vp_jfa_last_pass(self._h, ctypes.byref(frame), _vp(d_in), _vp(d_minus or None), _vp(d_plus or None),
                                     _vp(d_scratch), _vp(d_words), fill, _vp(d_sdf), algo))

    def jfa_can_start_from_mask(self, frame: Frame, algo: int = ALGO_TILED) -> bool:
        return bool(lib().vp_jfa_can_start_from_mask(ctypes.byref(frame), algo))

    def jfa_can_fuse_first_two(self, frame: Frame, algo: int = ALGO_TILED) -> bool:
        return bool(lib().vp_jfa_can_fuse_first_two(ctypes.byref(frame), algo))

    # -- id windows (vp_jfa_window_*): the slab form of the tile kernels; the layout inside a window is the library's
    def jfa_window_bytes(self, frame: Frame, planes: int) -> int:
        return int(lib().vp_jfa_window_bytes(ctypes.byref(frame), planes))

    def jfa_window_span(self, frame: Frame, planes: int, p0: int, p1: int):
        """[(byte offset, bytes), ...]: where the planes [p0, p1) of a window of `planes` planes live (one range, two above n = 1024)"""
        off, nb = (_sz * 2)(), (_sz * 2)()
        check(lib().vp_jfa_window_span(ctypes.byref(frame), planes, p0, p1, off, nb))
        return [(int(off[i]), int(nb[i])) for i in range(2) if nb[i]]

    def jfa_window_clear(self, frame: Frame, win: Window):
        check(lib().vp_jfa_window_clear(self._h, ctypes.byref(frame), ctypes.byref(win)))

    def jfa_window_init(self, frame: Frame, d_words: int, d_below, d_above, win: Window):
        check(lib().vp_jfa_window_init(self._h, ctypes.byref(frame), _vp(d_words), _vp(d_below or None), _vp(d_above or None), ctypes.byref(win)))

    def jfa_window_first_pass(self, frame: Frame, d_border_grid: int, win: Window):
        check(lib().vp_jfa_window_first_pass(self._h, ctypes.byref(frame), _vp(d_border_grid), ctypes.byref(win)))

    def jfa_window_first_two(self, frame: Frame, d_border_grid: int, win: Window):
        """passes n/2 and n/4 of a whole grid in one launch from its border mask"""
        check(lib().vp_jfa_window_first_two(self._h, ctypes.byref(frame), _vp(d_border_grid), ctypes.byref(win)))

    def jfa_window_pass(self, frame: Frame, k: int, win_in: Window, win_out: Window, stride=None):
        check(lib().vp_jfa_window_pass(self._h, ctypes.byref(frame), k, ctypes.byref(win_in), ctypes.byref(win_out), k if stride is None else stride))

    def jfa_window_last_pass(self, frame: Frame, win_in: Window, win_scratch: Window, d_words_region: int, fill: float, d_sdf_region: int, stride: int = 1):
        check(lib().vp_jfa_window_last_pass(self._h, ctypes.byref(frame), ctypes.byref(win_in), ctypes.byref(win_scratch), stride,
                                            _vp(d_words_region), fill, _vp(d_sdf_region)))

    # -- cyclic plane distribution (the first phase of the transposed multi-GPU pipeline)
    @staticmethod
    def jfa_cyclic_passes(frame: Frame, ranks: int) -> int:
        """passes of the sequence n/2, n/4, ... whose step is a multiple of `ranks`, counted from the first (0: the grid cannot be dealt cyclically)"""
        return int(lib().vp_jfa_cyclic_passes(ctypes.byref(frame), ranks))

    def jfa_window_first_two_cyclic(self, frame: Frame, d_border_grid: int, win: Window, ranks: int, rank: int):
        check(lib().vp_jfa_window_first_two_cyclic(self._h, ctypes.byref(frame), _vp(d_border_grid), ctypes.byref(win), ranks, rank))

    def jfa_window_pass_cyclic(self, frame: Frame, k: int, win_in: Window, win_out: Window, ranks: int, rank: int):
        check(lib().vp_jfa_window_pass_cyclic(self._h, ctypes.byref(frame), k, ctypes.byref(win_in), ctypes.byref(win_out), ranks, rank))

    def jfa_window_interleave(self, frame: Frame, win_in: Window, win_out: Window, ranks: int, count: int):
        check(lib().vp_jfa_window_interleave(self._h, ctypes.byref(frame), ctypes.byref(win_in), ctypes.byref(win_out), ranks, count))

    def surface(self, frame: Frame, d_words: int, d_below, d_above, d_border: int):
        check(lib().vp_surface(self._h, ctypes.byref(frame), _vp(d_words), _vp(d_below or None),
                               _vp(d_above or None), _vp(d_border)))

    def extract_count(self, frame: Frame, d_words: int, mode: int) -> int:
        n = ctypes.c_uint64()
        check(lib().vp_extract_count(self._h, ctypes.byref(frame), _vp(d_words), mode, ctypes.byref(n)))
        return int(n.value)

    def extract(self, frame: Frame, d_words: int, mode: int, d_sdf, d_records: int, d_values, capacity: int):
        check(lib().vp_extract(self._h, ctypes.byref(frame), _vp(d_words), mode, _vp(d_sdf or None), _vp(d_records),
                               _vp(d_values or None), capacity))

    # -- host-in / host-out (numpy arrays), the reference's Compute() convention
    def voxelize_host(self, frame: Frame, h_words, h_xyz, h_tri, algo: int = ALGO_TILED):
        check(lib().vp_voxelize_host(self._h, ctypes.byref(frame), h_words.ctypes.data_as(_vp),
                                     h_xyz.ctypes.data_as(_vp), h_xyz.shape[0], h_tri.ctypes.data_as(_vp),
                                     h_tri.shape[0], algo))

    def csg_host(self, h_a, h_b, op: int):
        check(lib().vp_csg_host(self._h, h_a.ctypes.data_as(_vp), h_b.ctypes.data_as(_vp), h_a.size, op))

    def jfa_host(self, frame: Frame, h_words, fill: float, h_sdf, algo: int = ALGO_TILED):
        check(lib().vp_jfa_host(self._h, ctypes.byref(frame), h_words.ctypes.data_as(_vp), fill,
                                h_sdf.ctypes.data_as(_vp), algo))

    # -- per-kernel device timing
    def prof_enable(self, on: bool = True):
        check(lib().vp_prof_enable(self._h, 1 if on else 0))

    def prof_select(self, names=None):
        """Time only the kernels whose timing keys are named (None = all): every event pair costs stream time."""
        mask = (1 << 64) - 1 if names is None else sum(1 << KERNELS.index(k) for k in names)
        check(lib().vp_prof_select(self._h, mask))

    def prof_reset(self):
        check(lib().vp_prof_reset(self._h))

    def prof(self):
        out = {}
        for i, name in enumerate(KERNELS):
            ms = ctypes.c_double()
            n = ctypes.c_uint64()
            check(lib().vp_prof_get(self._h, i, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[name] = {"ms": ms.value, "launches": int(n.value)}
        return out


class Multi:
    """vp_multi: one process, several devices (or several contexts on one device), Z-slabs.  Host arrays are numpy."""

    def __init__(self, devices):
        import numpy as np
        self._np = np
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        self._h = _vp()
        check(lib().vp_multi_create(devs, len(devices), ctypes.byref(self._h)))
        self.frame = None

    def close(self):
        if self._h:
            lib().vp_multi_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def count(self):
        return lib().vp_multi_count(self._h)

    def set_mesh(self, xyz, tri):
        np = self._np
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        tri = np.ascontiguousarray(tri, dtype=np.uint32)
        check(lib().vp_multi_set_mesh(self._h, xyz.ctypes.data, xyz.shape[0], tri.ctypes.data, tri.shape[0]))

    def voxelize(self, frame: Frame, algo=ALGO_TILED):
        check(lib().vp_multi_voxelize(self._h, ctypes.byref(frame), algo))
        self.frame = frame                                         # only once the driver has taken it

    def set_grid(self, frame: Frame, words):
        np = self._np
        words = np.ascontiguousarray(words, dtype=np.uint32)
        assert words.size == frame.words
        check(lib().vp_multi_set_grid(self._h, ctypes.byref(frame), words.ctypes.data))
        self.frame = frame

    def get_grid(self):
        out = self._np.empty(self.frame.words, dtype=self._np.uint32)
        check(lib().vp_multi_get_grid(self._h, out.ctypes.data))
        return out

    def csg(self, other, op):
        other = self._np.ascontiguousarray(other, dtype=self._np.uint32)
        check(lib().vp_multi_csg(self._h, other.ctypes.data, other.size, op))

    def jfa(self, fill=float("-inf"), algo=ALGO_TILED, mode=MULTI_HALO):
        check(lib().vp_multi_jfa(self._h, fill, algo, mode))

    def get_sdf(self):
        out = self._np.empty(self.frame.voxels, dtype=self._np.float32)
        check(lib().vp_multi_get_sdf(self._h, out.ctypes.data))
        return out

    def sync(self):
        check(lib().vp_multi_sync(self._h))

    @property
    def bytes_moved(self):
        return int(lib().vp_multi_bytes_moved(self._h))

    def window(self, rank: int):
        """(lo, hi, id_bytes): global planes the rank's id volumes covered during the last jfa, and the bytes of all its id buffers"""
        lo, hi, nb = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        check(lib().vp_multi_window(self._h, rank, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(nb)))
        return int(lo.value), int(hi.value), int(nb.value)
