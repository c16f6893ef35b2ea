#!/bin/bash
# dev helper (GPU box): time the JFA passes with every experimental library under tools/exp/
cd "$(dirname "$0")/.."
for lib in tools/exp/libvphip_*.so; do
  echo "=== $lib"
  VPHIP_LIB=$PWD/$lib python tools/jfa_passes.py "$@" 2>&1 | tail -12
done
