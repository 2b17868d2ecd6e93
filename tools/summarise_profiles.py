"""Build profiles/r01_pmc_summary.csv and profiles/traffic.json from the rocprofv3 output directories under gpurun_out/."""
import collections, csv, glob, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import shutil
newest = lambda pat: max(glob.glob(os.path.join(R, pat), recursive=True), key=os.path.getmtime)
def load(d):
    return list(csv.DictReader(open(newest(f"gpurun_out/{d}/**/*counter_collection.csv"))))
shutil.copy(newest("gpurun_out/r01_trace/**/*kernel_stats.csv"), os.path.join(R, "profiles/r01_kernel_stats.csv"))
def agg(rows):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        a[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(R, "profiles/r01_kernel_stats.csv")))}
fetch, write, mf = agg(load("r01_fetch")), agg(load("r01_write")), agg(load("r01_mfma"))
avg = lambda v: sum(v) / max(len(v), 1)
lines = ["# rocprofv3 PMC passes (separate runs: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES ...), bench.py --steps 20 --warmup 5 --no-overlap, MI355X, round 1",
         "# per-launch averages. FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) and is",
         "# (raw values in this table); k_bin_accum re-reads exactly what k_bin<2,true> wrote: its raw FETCH_SIZE is 0.50 x that WRITE_SIZE, so traffic.json doubles FETCH_SIZE.",
         "# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)  (MFMA pipe busy cycles per SIMD-cycle while the CU is busy)",
         "kernel,calls,avg_us,FETCH_SIZE_KiB,WRITE_SIZE_KiB,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_INSTS_MFMA,mfma_busy_frac"]
for k in sorted(stats, key=lambda k: -float(stats[k]["TotalDurationNs"]))[:14]:
    m = mf.get(k, {})
    mb, cu = avg(m.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), avg(m.get("SQ_BUSY_CU_CYCLES", [0]))
    lines.append(f'"{k[:64]}",{stats[k]["Calls"]},{float(stats[k]["AverageNs"])/1e3:.1f},{avg(fetch.get(k,{}).get("FETCH_SIZE",[0])):.0f},'
                 f'{avg(write.get(k,{}).get("WRITE_SIZE",[0])):.0f},{mb:.0f},{cu:.0f},{avg(m.get("SQ_INSTS_MFMA",[0])):.0f},{(mb/(4*cu) if cu else 0):.3f}')
open(os.path.join(R, "profiles/r01_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[4:]))
def per_dispatch(d, counter):
    rows = [r for r in load(d) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
out = {}
for grid, parity in (("color", 0), ("sdf", 1)):          # MapStep on one stream (--no-overlap) runs the colour branch first
    tot = 0.0
    # MI355X_MICROARCH.md (HBM / rocprofv3): on gfx950 FETCH_SIZE reports half of the bytes of a streaming read -> doubled.
    # Cross-check inside this function: k_bin_accum re-reads exactly the records k_bin<write> wrote; its raw FETCH_SIZE is
    # 0.50 x that kernel's WRITE_SIZE.  WRITE_SIZE is exact.
    for d, c, corr in (("r01_fetch", "FETCH_SIZE", 2.0), ("r01_write", "WRITE_SIZE", 1.0)):
        rows = per_dispatch(d, c)
        for kname in ("k_bin<2, false", "k_bin_colscan", "k_bin_scan", "k_bin<2, true", "k_bin_accum<2"):
            ks = [r for r in rows if kname in r["Kernel_Name"]][parity::2]
            tot += corr * sum(float(r["Counter_Value"]) for r in ks) / max(len(ks), 1) * 1024
    out[f"hashgrid_bwd_{grid}"] = tot
    print(grid, round(tot / 1e6, 1), "MB")
json.dump({**out, "_note": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the kernels of us_hashgrid_bwd_binned (k_bin_colscan, "
                           "k_bin_scan, k_bin<write>, k_bin_accum; the counting runs in the encoder's forward), rocprofv3 --pmc in separate passes, round 1; "
                           "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a streaming read; confirmed here: k_bin_accum's raw "
                           "FETCH_SIZE is 0.50 x the WRITE_SIZE of the records it re-reads)"},
          open(os.path.join(R, "profiles/traffic.json"), "w"), indent=1)
