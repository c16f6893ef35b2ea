"""The CMake consumer target: the reference is consumed as `add_subdirectory(vplib)` + `target_link_libraries(cli PRIVATE
vplib)` (/root/reference/vplib/CMakeLists.txt:20-30, apps/cli/CMakeLists.txt:22).  cuda_mesh_voxelization_amd/vplib/CMakeLists.txt exports
the same target name and options; this configures and builds tests/cpp/api_check.cpp and the CLI through it (no GPU needed to build)."""
import os
import shutil
import subprocess

import pytest

from cuda_mesh_voxelization_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_consumer_builds_against_the_vplib_target(tmp_path):
    build.build_lib()
    src = os.path.join(ROOT, "tests", "cpp", "cmake_consumer")
    vplib = os.path.join(ROOT, "cuda_mesh_voxelization_amd", "vplib")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    cfg = subprocess.run(["cmake", "-S", src, "-B", str(tmp_path), "-DVPLIB_SOURCE_DIR=" + vplib, "-DCMAKE_BUILD_TYPE=Release", "-DVPLIB_PROFILING=OFF"] + gen,
                         capture_output=True, text=True, timeout=600)
    assert cfg.returncode == 0, cfg.stdout[-3000:] + cfg.stderr[-3000:]
    bld = subprocess.run(["cmake", "--build", str(tmp_path), "-j", "4"], capture_output=True, text=True, timeout=1800)
    assert bld.returncode == 0, bld.stdout[-3000:] + bld.stderr[-3000:]
    exe, cli = os.path.join(str(tmp_path), "api_check"), os.path.join(str(tmp_path), "cli", "cli")
    assert os.path.exists(exe) and os.path.exists(cli)
    # linked against the HIP library through the target, found at run time through the rpath the target carries
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libvphip.so" in ldd and "not found" not in [ln for ln in ldd.splitlines() if "libvphip" in ln][0]
    # the options reach the consumer's compile line as the reference's definitions
    cc = open(os.path.join(str(tmp_path), "build.ninja")).read() if gen else ""
    assert (not gen) or ("-DPROFILING=0" in cc and "-DLOGGING=1" in cc and "-ffp-contract=off" in cc)
    # the CLI built this way is the CLI: the CPU type runs without a GPU
    out = subprocess.run([cli, os.path.join(ROOT, "assets", "d20.obj"), "-n", "32", "-t", "0"], capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
