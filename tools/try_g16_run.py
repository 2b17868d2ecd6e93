import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import unislam_amd as us
from g15_settings import G16
import test_gpu_slam as T
g = dict(np.load("/root/repo/tests/golden/g16_policy.npz"))
for k, nf in ((16, 9), (12, 9)):
    slam, draws = T._resume(us, g, G16, k)
    try:
        slam.run(n_frames=k + nf, start=k, total=G16["n_frames"])
    except AssertionError as e:
        print("stopped:", str(e)[:200])
    for f in range(k, k + nf):
        if f not in slam.history["track_iters"]: break
        est, ref = slam.estimate_c2w_list[f].cpu().numpy(), g["est_c2w"][f]
        print(f, "iters", slam.history["track_iters"][f], int(g["track_iters"][f]), "tb", int(slam.history["tracking_back"].get(f, -1)), int(g["tracking_back"][f]),
              "dt %.1e" % np.abs(est[:3, 3] - ref[:3, 3]).max())
