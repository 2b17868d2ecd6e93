#!/bin/bash
# development: kernel variants by -D flags (on the GPU box): [CMD="python tools/time_render.py"] tools/acc_variants.sh "<flags>" ...
# (the garbage-result timing variants J_WR_X_* / MLP_SKIP_SETUP need -DUS_EXPERIMENTS in the flags as well: include/unislam_hip_experiments.h)
# default command: the per-kernel table of a short bench run, table-gradient kernels
for flags in "$@"; do
  touch uni-slam_amd/csrc/*.hip; make -s -C uni-slam_amd/csrc EXTRA="$flags" 2>&1 | grep -v "warning\|^ \|\^\|generated\|In file" | head -3
  echo "=== [$flags]"
  if [ -n "$CMD" ]; then $CMD 2>&1 | grep -v "amdgpu.ids"; else tools/kprof.sh accv 2>&1 | grep "k_j"; fi
done
touch uni-slam_amd/csrc/*.hip; make -s -C uni-slam_amd/csrc 2>&1 | grep -v "warning\|^ \|\^\|generated\|In file"
