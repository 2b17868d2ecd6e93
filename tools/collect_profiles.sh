#!/bin/bash
# tools/collect_profiles.sh <tag> -- run on the GPU box (gpurun): the round's bench line, the rocprofv3 kernel statistics and the PMC
# passes (each counter set in its own run, --kernel-trace only) that tools/summarise_profiles.py <tag> turns into profiles/<tag>_*;
# from round 3 on also the traces of the replayed mapping iteration (timeline), of the joint_opt window and of the tracking iteration.
# rocprofv3 gets the program itself after `--` (no env / bash -c hop).
R=$PWD; T=${1:-r03}
mkdir -p $R/gpurun_out
timeout -k 10 900 python3 bench.py > $R/gpurun_out/bench_$T.json 2> $R/gpurun_out/bench_$T.err || { tail -5 $R/gpurun_out/bench_$T.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tracking --no-extras --no-overlap"
for d in trace fetch write mfma sq map window track; do rm -rf $R/gpurun_out/${T}_$d; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_trace -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/${T}_trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/${T}_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/${T}_write.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/${T}_mfma -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/${T}_mfma.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/${T}_sq -o runc -- python3 $R/bench.py $ARGS > $R/gpurun_out/${T}_sq.log 2>&1 || exit 6
# the replayed graph with its side streams (timeline), the joint_opt window, the tracking iteration
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_map -o runc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tracking --no-extras --no-probe > $R/gpurun_out/${T}_map.log 2>&1 || exit 7
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_window -o runc -- python3 $R/tools/time_window.py 16 256 0 0 > $R/gpurun_out/${T}_window.log 2>&1 || exit 8
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_track -o runc -- python3 $R/tools/prof_track_fused.py > $R/gpurun_out/${T}_track.log 2>&1 || exit 9
cd $R && cut -c1-300 gpurun_out/bench_$T.json && find gpurun_out/${T}_trace gpurun_out/${T}_fetch -name "*.csv" | head
