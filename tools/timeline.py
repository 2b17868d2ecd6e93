"""python tools/timeline.py <rocprofv3 output dir> [iterations back] [delimiter kernel, default k_adam_segs]: the kernels of one
iteration (between two Adam passes -- or two launches of the given kernel, e.g. k_pose_window_step for a tracking trace) of a traced
run, in start order: start (us after the previous Adam pass ended), duration, idle time before it (no kernel of the trace
running), queue, name."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
delim = sys.argv[3] if len(sys.argv) > 3 else "k_adam_segs"
idx = [i for i, r in enumerate(rows) if delim in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]["End_Timestamp"]); busy_until = t0; idle = 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - busy_until); idle += gap
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  idle {gap / 1e3:5.1f}  q{r['Queue_Id']:>2s} {r['Kernel_Name'][:70]}")
    busy_until = max(busy_until, e)
print(f"iteration {(busy_until - t0) / 1e3:.1f} us, of which no kernel running {idle / 1e3:.1f} us")
