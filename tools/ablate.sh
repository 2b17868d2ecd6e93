#!/bin/bash
# tools/ablate.sh "<-D flags>" ... : build tools/bwd_bench.hip once per flag set and print the per-kernel rocprofv3 averages
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics $flags $R/tools/bwd_bench.hip -o /tmp/bb_$i 2>/dev/null || { echo "build failed: $flags"; continue; }
  rm -rf /tmp/abl_$i
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$i -o bb -- /tmp/bb_$i > /tmp/abl_$i.log 2>&1
  echo "=== flags: [$flags]"
  grep BINNED /tmp/abl_$i.log | grep level-major
  python3 $R/tools/kstats.py /tmp/abl_$i | grep -v sliced | grep -v fillBuffer
done
