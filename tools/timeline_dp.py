"""python tools/timeline_dp.py <rocprofv3 output dir of tools/time_dp_rank.py>: the kernels of the last data-parallel iteration of the trace (the
last stretch between two sampling launches with the two-grid kernels, poses fixed, whose optimiser step runs in two parts), in start order: start, duration, idle before, queue, name."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ws = [i for i, r in enumerate(rows) if "k_window_sample" in r["Kernel_Name"]]
sel = None
for a, b in zip(ws[:-1], ws[1:]):
    names = [r["Kernel_Name"] for r in rows[a:b]]
    if sum(1 for n in names if "k_adam_segs<" in n) == 2 and any("k_jwrite" in n for n in names) and any("k_jfwd<true, true, false>" in n for n in names) and b - a < 40:
        sel = (a, b)
a, b = sel
t0 = int(rows[a]["Start_Timestamp"]); busy = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q, name = r["Queue_Id"], r["Kernel_Name"][:64]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  idle {max(0, s - busy) / 1e3:5.1f}  q{q:>2s} {name}")
    busy = max(busy, e)
print(f"span {(busy - t0) / 1e3:.1f} us")
