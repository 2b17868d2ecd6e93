#!/bin/bash
# tools/pmc_mlp.sh [out file]: the decoder pair's attribution pass -- rocprofv3 --pmc passes (each counter set in its own run,
# --kernel-trace only) over tools/time_mlp_pair.py (262 144 points, both 2 x 32 decoders, split bf16), per-kernel averages of
# k_mlp_fwd_pair / k_mlp_bwd_pair.  Run on the GPU box: bash tools/pmc_mlp.sh gpurun_out/r06_mlp_pmc.txt
R=$PWD; OUT=${1:-$R/gpurun_out/r06_mlp_pmc.txt}; case "$OUT" in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU" \
           "SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32"; do
  rm -rf /tmp/pmc_out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_out -o p -- python3 $R/tools/time_mlp_pair.py > /tmp/pmc.log 2>&1
  echo "== $SET" >> $OUT
  python3 - >> $OUT <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True)
if not f:
    print("   (no counters: " + open("/tmp/pmc.log").read()[-300:].replace("\n", " | ") + ")")
else:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    for k in sorted(acc):
        if "mlp" in k:
            print("  ", k, "calls", cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
done
cat $OUT
