#!/bin/bash
# development: kernel variants by -D flags (on the GPU box): tools/acc_variants.sh "<flags>" ...
for flags in "$@"; do
  touch uni-slam_amd/csrc/*.hip; make -s -C uni-slam_amd/csrc EXTRA="$flags" 2>&1 | grep -v warning | head -3
  echo "=== [$flags]"; tools/kprof.sh accv 2>&1 | grep "k_j"
done
touch uni-slam_amd/csrc/*.hip; make -s -C uni-slam_amd/csrc
