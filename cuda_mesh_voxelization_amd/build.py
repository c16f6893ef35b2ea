"""Build recipes for the native parts (in-tree, so the built files travel with the repo snapshot).

  libvphip.so   HIP kernels + C ABI (include/vphip.h), hipcc --offload-arch=gfx950
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

HIP_SOURCES = ["capi.hip", "vox.hip", "csg.hip", "jfa.hip", "extract.hip", "multi.hip"]
JFA_PARTS = 5                    # jfa.hip is compiled as parts 0 .. 4 side by side (-DVP_JFA_PART=i, see the top of the file)
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


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "vp_internal.h"), os.path.join(ROOT, "include", "vphip.h")]
    if force or _newer(LIB, deps):
        # one hipcc per source, side by side (jfa.hip alone is ~2 minutes of template instantiations), then one link
        from concurrent.futures import ThreadPoolExecutor
        objdir = os.path.join(PKG, "build")
        os.makedirs(objdir, exist_ok=True)
        flags = [f for f in HIP_FLAGS if f != "-shared"]
        objs, cmds = [], []
        for s_ in srcs:
            if os.path.basename(s_) == "jfa.hip":               # the longest first: five parts of the tile-kernel instantiations
                for part in range(JFA_PARTS):
                    o = os.path.join(objdir, "jfa.hip.part%d.o" % part)
                    objs.append(o)
                    cmds.insert(part, [_hipcc()] + flags + ["-DVP_JFA_PART=%d" % part, "-c", s_, "-o", o])
            else:
                o = os.path.join(objdir, os.path.basename(s_) + ".o")
                objs.append(o)
                cmds.append([_hipcc()] + flags + ["-c", s_, "-o", o])
        if verbose:
            for c in cmds:
                print(" ".join(c))
        with ThreadPoolExecutor(max_workers=min(max(2, (os.cpu_count() or 4) - 1), len(cmds))) as ex:
            list(ex.map(subprocess.check_call, cmds))
        link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
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
