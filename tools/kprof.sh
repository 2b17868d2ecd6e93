#!/bin/bash
# tools/kprof.sh <tag> [bench.py args]: on the GPU box, rocprofv3 --kernel-trace --stats of a short bench.py run -> gpurun_out/<tag>_trace,
# per-kernel table on stdout.  (rocprofv3 gets the program itself after `--`: no env / bash -c hop.)
R=$PWD; tag=$1; shift
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/${tag}_trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o runc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tracking --no-extras --no-overlap "$@" > $R/gpurun_out/${tag}_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_trace.log; exit 2; }
cd $R && python3 tools/kstats.py gpurun_out/${tag}_trace | sort -t' ' -k1,1 | grep -v "at::native\|rocclr" 
