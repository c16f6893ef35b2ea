"""Build recipes for the native parts (in-tree, so the built files travel with the repo snapshot).

  libvphip.so   HIP kernels + C ABI (include/vphip.h), hipcc --offload-arch=gfx950
  libvphip_hooks.so   the same with the test hooks of vox.hip / multi.hip compiled in (-DVP_TEST_HOOKS); only tests load it
  vpcli         C++23 CLI mirroring the reference's apps/cli (links libvphip.so)

hipcc cross-compiles gfx950 code objects without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libvphip.so")
CLI = os.path.join(PKG, "vpcli")

HIP_SOURCES = ["capi.hip", "vox.hip", "csg.hip", "jfa_seed.hip", "jfa_first_two.hip", "jfa_dense.hip", "extract.hip", "multi.hip"]
DENSE_PARTS = 10                 # jfa_dense.hip is compiled once per id format and pass kind, side by side (-DVP_DENSE_PART=1..10, see the end of the file)
HOOK_SOURCES = ["vox.hip", "multi.hip"]     # the sources that read test hooks from the environment under -DVP_TEST_HOOKS (libvphip_hooks.so)
HOOKS_LIB = os.path.join(PKG, "libvphip_hooks.so")
# -ffp-contract=off is part of the parity contract: an FMA changes the bitmask / sdf bits.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libvphip.so")
    return exe


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _jobs(n: int) -> int:
    """hipcc processes side by side: at most 8 and never more than cores - 1 (each holds ~1-2 GB while it instantiates the tile kernels);
    VP_BUILD_JOBS overrides"""
    env = os.environ.get("VP_BUILD_JOBS")
    if env:
        return max(1, min(int(env), n))
    return max(1, min(8, max(2, (os.cpu_count() or 4) - 1), n))


def build_lib(force: bool = False, verbose: bool = False, hooks: bool = False) -> str:
    """libvphip.so; hooks=True: also libvphip_hooks.so -- the same objects except vox.hip / multi.hip compiled with -DVP_TEST_HOOKS, the
    build the tests that force rare paths load (the default library reads no environment variable on any call path)."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "vp_internal.h"), os.path.join(CSRC, "jfa_common.h"), os.path.join(ROOT, "include", "vphip.h")]
    objdir = os.path.join(PKG, "build")
    flags = [f for f in HIP_FLAGS if f != "-shared"]

    def link(objs, out):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    objs = []
    for s_ in srcs:
        base = os.path.basename(s_)
        if base == "jfa_dense.hip":
            objs += [os.path.join(objdir, "jfa_dense.hip.part%d.o" % part) for part in range(1, DENSE_PARTS + 1)]
        else:
            objs.append(os.path.join(objdir, base + ".o"))
    # (the objects may be gone while the library is current -- a shipped .so, a cleaned build/: the hooks library below links them, so a
    # missing one forces the compile step)
    if force or _newer(LIB, deps) or (hooks and (force or _newer(HOOKS_LIB, deps + [LIB])) and not all(os.path.exists(o) for o in objs)):
        # one hipcc per source, side by side (the tile kernel alone is over a minute of template instantiations per id format), then one link
        from concurrent.futures import ThreadPoolExecutor
        os.makedirs(objdir, exist_ok=True)
        cmds = []
        for s_ in srcs:
            base = os.path.basename(s_)
            if base == "jfa_dense.hip":                         # the longest first
                for part in range(1, DENSE_PARTS + 1):
                    cmds.insert(part - 1, [_hipcc()] + flags + ["-DVP_DENSE_PART=%d" % part, "-c", s_, "-o", os.path.join(objdir, "jfa_dense.hip.part%d.o" % part)])
            else:
                cmds.append([_hipcc()] + flags + ["-c", s_, "-o", os.path.join(objdir, base + ".o")])
        if verbose:
            for c in cmds:
                print(" ".join(c))
        with ThreadPoolExecutor(max_workers=_jobs(len(cmds))) as ex:
            list(ex.map(subprocess.check_call, cmds))
        link(objs, LIB)
    if hooks and (force or _newer(HOOKS_LIB, deps + [LIB])):
        hobjs = list(objs)
        for name in HOOK_SOURCES:
            o = os.path.join(objdir, name + ".hooks.o")
            cmd = [_hipcc()] + flags + ["-DVP_TEST_HOOKS", "-c", os.path.join(CSRC, name), "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            hobjs[hobjs.index(os.path.join(objdir, name + ".o"))] = o
        link(hobjs, HOOKS_LIB)
    return LIB


def build_cli(force: bool = False, verbose: bool = False) -> str:
    vplib = os.path.join(PKG, "vplib")
    src = os.path.join(PKG, "apps", "cli", "main.cpp")
    if not os.path.exists(src):
        raise RuntimeError("CLI sources missing")
    deps = [src] + [os.path.join(dp, f) for dp, _, fs in os.walk(vplib) for f in fs]
    build_lib(force=False, verbose=verbose)
    if force or _newer(CLI, deps + [LIB]):
        cpps = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(vplib, "src")) for f in fs if f.endswith(".cpp")]
        cmd = ["g++", "-std=c++23", "-O2", "-ffp-contract=off", "-fopenmp", "-DPROFILING=1",
               "-I", os.path.join(vplib, "include"), "-I", os.path.join(ROOT, "include"),
               src] + cpps + ["-o", CLI, "-L", PKG, "-lvphip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
