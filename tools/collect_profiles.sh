#!/bin/bash
# tools/collect_profiles.sh -- run on the GPU box (gpurun): the round's bench line, the rocprofv3 kernel statistics and the PMC
# passes (each counter set in its own run, --kernel-trace only) that tools/summarise_profiles.py turns into profiles/*.
R=$PWD
mkdir -p $R/gpurun_out
timeout -k 10 600 python3 bench.py > $R/gpurun_out/bench_r01.json 2> $R/gpurun_out/bench_r01.err || { tail -5 $R/gpurun_out/bench_r01.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tracking --no-overlap"
rm -rf $R/gpurun_out/r01_trace $R/gpurun_out/r01_fetch $R/gpurun_out/r01_write $R/gpurun_out/r01_mfma
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01_trace -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/r01_trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01_fetch -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/r01_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01_write -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/r01_write.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/r01_mfma -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/r01_mfma.log 2>&1 || exit 5
cd $R && cat gpurun_out/bench_r01.json | cut -c1-400 && find gpurun_out/r01_trace gpurun_out/r01_fetch -name "*.csv" | head
