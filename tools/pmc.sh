#!/bin/bash
# tools/pmc.sh "<counters>" [hipcc -D flags] : rocprofv3 --pmc pass over tools/bwd_bench.hip, per-kernel averages of the counters
R=$PWD
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics $2 $R/tools/bwd_bench.hip -o /tmp/bb_pmc 2>/dev/null || { echo "build failed"; exit 1; }
rm -rf /tmp/pmc_out
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/pmc_out -o p -- /tmp/bb_pmc > /tmp/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True)
if not f:
    print(open("/tmp/pmc.log").read()[-2000:]); raise SystemExit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    if "sliced" in k: continue
    print(k, "calls", cnt[k], {c: round(v / cnt[k], 1) for c, v in acc[k].items()})
PY
