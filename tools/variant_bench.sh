#!/bin/bash
# tools/variant_bench.sh "<EXTRA flags>" ... : rebuild the library with each flag set (on the GPU box) and print bench.py's kernel_ms
for flags in "$@"; do
  touch uni-slam_amd/csrc/*.hip
  make -C uni-slam_amd/csrc EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "=== EXTRA: [$flags]"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-tracking | python -c "
import json,sys;r=json.loads(sys.stdin.read());print(round(r['ms_per_step'],4), r['kernel_ms']['hashgrid_bwd_color'], r['kernel_ms']['hashgrid_bwd_sdf'], round(r['bf16_decoders']['ms_per_step'],4))"
done
