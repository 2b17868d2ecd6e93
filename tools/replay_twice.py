"""Is the per-frame difference between the HIP replay of fixture g15 and the reference's loop a parity defect or the amplification of rounding
differences?  Run the SAME HIP code on the SAME draws twice: the two runs differ only by the order of float atomics / f64 LDS
sums inside the table gradient (a last-bit difference in a handful of entries per iteration), and are compared with each other exactly as
each is compared with the reference.   python tools/replay_twice.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import unislam_amd as us
from g15_settings import G15
import test_gpu_slam as T

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g15_sequence.npz")))
runs = []
for k in range(2):
    slam, dev = T._replay_against(us, g, G15)
    runs.append((slam.estimate_c2w_list[:, :3, 3].cpu().double().numpy(), np.array(slam.history["losses"]), dev.numpy()))
a, b = runs
rel = np.abs(a[1] - b[1]) / np.abs(b[1])
first = np.nonzero(rel > 0)[0]
print(f"HIP run 1 against HIP run 2 (same draws): the first iteration whose loss differs is {int(first[0]) if len(first) else None}; relative loss difference at "
      f"iterations 1, 2, 4, 8, 16, 32, 50: {[float('%.1e' % rel[i]) for i in (0, 1, 3, 7, 15, 31, 49)]}")
print("per-frame |t_1 - t_2| (mm):", np.array2string(1e3 * np.linalg.norm(a[0] - b[0], axis=1), precision=2, max_line_width=220))
print("per-frame |t_1 - t_ref| (mm):", np.array2string(1e3 * a[2], precision=2, max_line_width=220))
print("per-frame |t_2 - t_ref| (mm):", np.array2string(1e3 * b[2], precision=2, max_line_width=220))
