"""python tools/summarise_profiles.py <tag>: build profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.csv, profiles/<tag>_sq_stalls.txt,
profiles/<tag>_bench.json and profiles/traffic.json from the rocprofv3 output directories tools/collect_profiles.sh left under gpurun_out/."""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = sys.argv[1] if len(sys.argv) > 1 else "r03"
newest = lambda pat: max(glob.glob(os.path.join(R, pat), recursive=True), key=os.path.getmtime)
def load(d):
    return list(csv.DictReader(open(newest(f"gpurun_out/{T}_{d}/**/*counter_collection.csv"))))
shutil.copy(newest(f"gpurun_out/{T}_trace/**/*kernel_stats.csv"), os.path.join(R, f"profiles/{T}_kernel_stats.csv"))
if os.path.exists(os.path.join(R, f"gpurun_out/bench_{T}.json")):
    shutil.copy(os.path.join(R, f"gpurun_out/bench_{T}.json"), os.path.join(R, f"profiles/{T}_bench.json"))
# round 3: the joint_opt window and the tracking iteration (kernel statistics), timelines of one replayed mapping / window / tracking iteration
import subprocess
for tag, delim in (("window", "k_adam_segs"), ("track", "k_pose_window_step"), ("map", "k_adam_segs")):
    try:
        if tag != "map":
            shutil.copy(newest(f"gpurun_out/{T}_{tag}/**/*kernel_stats.csv"), os.path.join(R, f"profiles/{T}_{tag}_kernel_stats.csv"))
        txt = subprocess.run([sys.executable, os.path.join(R, "tools/timeline.py"), os.path.join(R, f"gpurun_out/{T}_{tag}"), "3", delim],
                             capture_output=True, text=True).stdout
        name = f"profiles/{T}_timeline.txt" if tag == "map" else f"profiles/{T}_{tag}_timeline.txt"
        what = {"map": "the replayed mapping iteration (bench.py headline: MapWindow graph, joint_opt off)", "window": "the joint_opt mapping iteration (tools/time_window.py 16 256)",
                "track": "the tracking iteration (tools/prof_track_fused.py: TrackStep.iterate_fused, 2000 x 40)"}[tag]
        open(os.path.join(R, name), "w").write(f"# tools/timeline.py on a rocprofv3 --kernel-trace of {what}, MI355X, {T}: start (us), +duration, idle before, queue, kernel\n" + txt)
    except Exception as e:
        print("no", tag, "trace:", e)


def agg(rows):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        a[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(R, f"profiles/{T}_kernel_stats.csv")))}
fetch, write, mf = agg(load("fetch")), agg(load("write")), agg(load("mfma"))
avg = lambda v: sum(v) / max(len(v), 1)
# the first launches of a run are not the steady state (set-up iterations on placeholder rays): per-kernel MEDIANS for the byte counters
med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
lines = [f"# rocprofv3 PMC passes (separate runs: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA), bench.py --steps 20 --warmup 5 --no-overlap, MI355X, {T}",
         "# (--no-overlap: every launch on ONE stream, so that a counter row belongs to one kernel -- the optimiser then shows as k_adam_segs<4> + k_mlp_reduce_pair + k_step_inc,",
         "#  where the replayed headline runs ONE k_adam_segs_model<4> launch (profiles/*_timeline.txt); same bytes, same kernels otherwise)",
         "# per-launch medians of the byte counters, averages of the rest. FETCH_SIZE / WRITE_SIZE in KiB as reported (raw): on gfx950 FETCH_SIZE reports half of the bytes of a wide",
         "# coalesced streaming read (MI355X_MICROARCH.md, HBM); traffic.json doubles it for the kernels whose reads are such streams.",
         "# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)  (MFMA pipe busy cycles per SIMD-cycle while the CU is busy)",
         "kernel,calls,avg_us,FETCH_SIZE_KiB,WRITE_SIZE_KiB,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_INSTS_MFMA,mfma_busy_frac"]
for k in sorted(stats, key=lambda k: -float(stats[k]["TotalDurationNs"]))[:16]:
    m = mf.get(k, {})
    mb, cu = avg(m.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), avg(m.get("SQ_BUSY_CU_CYCLES", [0]))
    lines.append(f'"{k[:64]}",{stats[k]["Calls"]},{float(stats[k]["AverageNs"])/1e3:.1f},{med(fetch.get(k,{}).get("FETCH_SIZE",[0])):.0f},'
                 f'{med(write.get(k,{}).get("WRITE_SIZE",[0])):.0f},{mb:.0f},{cu:.0f},{avg(m.get("SQ_INSTS_MFMA",[0])):.0f},{(mb/(4*cu) if cu else 0):.3f}')
open(os.path.join(R, f"profiles/{T}_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[4:]))
try:
    sq = agg(load("sq"))
    out = [f"# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU, bench.py --steps 20 --warmup 5 --no-overlap, MI355X, {T}",
           "# per-launch averages; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over waves; SQ_LDS_* in cycles summed over CUs"]
    for k in sorted(sq, key=lambda k: -float(stats.get(k, {"TotalDurationNs": 0})["TotalDurationNs"]))[:12]:
        v = {c: avg(x) for c, x in sq[k].items()}
        wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
        out.append(f'{k[:60]:60s} wait_any {v.get("SQ_WAIT_ANY", 0) / wc:.2f} wait_inst {v.get("SQ_WAIT_INST_ANY", 0) / wc:.2f} active {v.get("SQ_ACTIVE_INST_ANY", 0) / wc:.2f} of wave cycles; '
                   f'lds_active {v.get("SQ_LDS_IDX_ACTIVE", 0):.0f} lds_bank_conflict {v.get("SQ_LDS_BANK_CONFLICT", 0):.0f} insts_valu {v.get("SQ_INSTS_VALU", 0):.0f}')
    open(os.path.join(R, f"profiles/{T}_sq_stalls.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[2:]))
except Exception as e:
    print("no SQ pass:", e)
# ---- HBM traffic per launch of the table-gradient entry points (bench.py's roofline.traffic)
# FETCH_SIZE x 2: the record streams are wide coalesced reads (gfx950 counts their 128-byte requests as 64 bytes); WRITE_SIZE is exact.
def traffic(names):
    tot = 0.0
    for d, c, corr in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        a = agg(load(d))
        for k in a:
            if any(n in k for n in names):
                tot += corr * med(a[k][c]) * 1024
    return tot
out = {"hashgrid_bwd_joint": traffic(("k_jcolscan", "k_jscan", "k_jitems", "k_jwrite", "k_jaccum")), "hashgrid_fwd_joint": traffic(("k_jfwd<true, true, false>",))}
for k, v in out.items():
    print(k, round(v / 1e6, 1), "MB")
json.dump({**out, "_note": f"bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, per-kernel medians summed over the kernels of the entry point (hashgrid_bwd_joint: "
                           f"k_jcolscan, k_jscan, k_jitems, k_jwrite, k_jaccum; the counting runs in the encoder), rocprofv3 --pmc in separate passes, {T}; FETCH_SIZE doubled per "
                           "MI355X_MICROARCH.md (gfx950 reports half of a wide streaming read)"},
          open(os.path.join(R, "profiles/traffic.json"), "w"), indent=1)
