#!/bin/bash
# dev helper: rebuild libvphip.so from anywhere
cd "$(dirname "$0")/.." && python -c "from cuda_mesh_voxelization_amd import build; build.build_lib(force=True)"
