#!/bin/bash
# dev helper: build an experimental variant of libvphip.so into tools/exp/ (git-ignored), e.g.
#   tools/exp_build.sh w5 -DVP_EXP_WAVES=5      then   VPHIP_LIB=tools/exp/libvphip_w5.so python tools/jfa_passes.py
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude "$@" \
    cuda_mesh_voxelization_amd/csrc/{capi,vox,csg,jfa,extract,multi}.hip -o tools/exp/libvphip_$name.so
echo tools/exp/libvphip_$name.so
