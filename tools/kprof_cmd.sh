#!/bin/bash
# tools/kprof_cmd.sh <tag> <python script> [args]: rocprofv3 --kernel-trace --stats of a python script on the GPU box, per-kernel table on stdout
R=$PWD; tag=$1; shift
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/${tag}_trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -o runc -- python3 $R/"$@" > $R/gpurun_out/${tag}_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_trace.log; exit 2; }
cd $R && python3 tools/kstats.py gpurun_out/${tag}_trace | sort -t' ' -k1,1 | grep -v "at::native\|rocclr"
