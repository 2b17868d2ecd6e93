#!/bin/bash
# tools/pmc_py.sh "<counters>" script.py [args] : rocprofv3 --pmc pass over a python script, per-kernel averages of the counters
R=$PWD
C="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_out -o p -- python3 $R/"$@" > /tmp/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True)
if not f:
    print(open("/tmp/pmc.log").read()[-1500:]); raise SystemExit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    if "mlp" in k or "k_bin" in k or "k_fwd" in k:
        print(k, "calls", cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
