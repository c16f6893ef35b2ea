#!/bin/bash
# dev helper: build an experimental variant of libvphip.so into tools/exp/ (git-ignored), e.g.
#   tools/exp_build.sh w5 -DVP_EXP_WAVES=5      then   VPHIP_LIB=tools/exp/libvphip_w5.so python tools/ab_step.py ...
# The -D flags of a variant reach the three JFA sources (jfa_seed.hip, jfa_first_two.hip, jfa_dense.hip -- the latter as its ten
# build parts, side by side); the other sources are compiled once into tools/exp/obj/ and shared by all variants (rebuilt when a
# source or header is newer).  Do not edit sources while a build runs.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/exp/obj
C=cuda_mesh_voxelization_amd/csrc
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude"
(
  flock 9
  for s in capi vox csg extract multi; do
    o=tools/exp/obj/$s.o
    if [ ! -f $o ] || [ $C/$s.hip -nt $o ] || [ $C/vp_internal.h -nt $o ] || [ include/vphip.h -nt $o ]; then
      $CC -c $C/$s.hip -o $o &
    fi
  done
  wait
) 9> tools/exp/obj/.lock
pids=""
for p in 1 2 3 4 5 6 7 8 9 10; do
  $CC "$@" -DVP_DENSE_PART=$p -c $C/jfa_dense.hip -o tools/exp/obj/jfa_dense_${name}_$p.o &
  pids="$pids $!"
done
for s in jfa_seed jfa_first_two; do
  $CC "$@" -c $C/$s.hip -o tools/exp/obj/${s}_${name}.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/exp/obj/{capi,vox,csg,extract,multi}.o tools/exp/obj/jfa_dense_${name}_{1,2,3,4,5,6,7,8,9,10}.o \
    tools/exp/obj/jfa_seed_${name}.o tools/exp/obj/jfa_first_two_${name}.o -o tools/exp/libvphip_$name.so
rm -f tools/exp/obj/jfa_dense_${name}_{1,2,3,4,5,6,7,8,9,10}.o tools/exp/obj/jfa_seed_${name}.o tools/exp/obj/jfa_first_two_${name}.o
echo tools/exp/libvphip_$name.so
