"""How far does a replay that starts from a snapshot of fixture g16 (the reference loop's whole state at frame k) stay on the reference's
decisions and poses?  (tests/test_gpu_slam.py holds the first frames.)   python tools/try_g16_run.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import unislam_amd as us
from g15_settings import G16
import test_gpu_slam as T

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g16_policy.npz")))
for k, nf in ((16, 9), (12, 9)):
    slam, draws = T._resume(us, g, G16, k)
    try:
        slam.run(n_frames=k + nf, start=k, total=G16["n_frames"])
    except AssertionError as e:
        print("left the reference's draw stream:", str(e)[:200])
    for f in range(k, k + nf):
        if f not in slam.history["track_iters"]:
            break
        est, ref = slam.estimate_c2w_list[f].cpu().numpy(), g["est_c2w"][f]
        print(f"from {k}: frame {f}: tracking iterations {slam.history['track_iters'][f]} (reference {int(g['track_iters'][f])}), tracking back "
              f"{int(slam.history['tracking_back'].get(f, -1))} ({int(g['tracking_back'][f])}), |t - t_ref| {np.abs(est[:3, 3] - ref[:3, 3]).max():.1e} m")
