#!/bin/bash
# tools/variant_mlp.sh "<EXTRA flags>" ... : rebuild the library with each flag set (on the GPU box) and print tools/time_mlp.py
for flags in "$@"; do
  touch uni-slam_amd/csrc/mlp.hip
  make -C uni-slam_amd/csrc EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "=== EXTRA: [$flags]"
  timeout -k 10 100 python tools/time_mlp.py | grep "width 32 hidden 2"
done
