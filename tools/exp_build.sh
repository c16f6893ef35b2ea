#!/bin/bash
# dev helper: build an experimental variant of libvphip.so into tools/exp/ (git-ignored), e.g.
#   tools/exp_build.sh w5 -DVP_EXP_WAVES=5      then   VPHIP_LIB=tools/exp/libvphip_w5.so python tools/jfa_passes.py
# The -D flags of a variant only reach jfa.hip (where every experiment macro lives); the other five sources are compiled once into
# tools/exp/obj/ and shared by all variants (rebuilt when a source or header is newer).  jfa.hip is compiled as its five build parts
# side by side (-DVP_JFA_PART=0..4).  Do not edit sources while a build runs.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/exp/obj
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude"
(
  flock 9
  for s in capi vox csg extract multi; do
    o=tools/exp/obj/$s.o
    if [ ! -f $o ] || [ cuda_mesh_voxelization_amd/csrc/$s.hip -nt $o ] || [ cuda_mesh_voxelization_amd/csrc/vp_internal.h -nt $o ] || [ include/vphip.h -nt $o ]; then
      $CC -c cuda_mesh_voxelization_amd/csrc/$s.hip -o $o &
    fi
  done
  wait
) 9> tools/exp/obj/.lock
pids=""
for p in 0 1 2 3 4; do
  $CC "$@" -DVP_JFA_PART=$p -c cuda_mesh_voxelization_amd/csrc/jfa.hip -o tools/exp/obj/jfa_${name}_$p.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/exp/obj/{capi,vox,csg,extract,multi}.o tools/exp/obj/jfa_${name}_{0,1,2,3,4}.o -o tools/exp/libvphip_$name.so
rm -f tools/exp/obj/jfa_${name}_{0,1,2,3,4}.o
echo tools/exp/libvphip_$name.so
