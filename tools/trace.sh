#!/bin/bash
# tools/trace.sh <tag> <script.py> [args] -- on the GPU box: rocprofv3 --kernel-trace --stats of a python script (the program itself after
# `--`), csv under gpurun_out/<tag>/, then the per-kernel table (tools/kstats.py)
R=$PWD; T=$1; shift
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T -o run -- python3 $R/"$@" > $R/gpurun_out/$T.log 2>&1 || { tail -20 $R/gpurun_out/$T.log; exit 2; }
cd $R && python3 tools/kstats.py gpurun_out/$T | sort -t_ -k3 | head -60
