"""development: the HIP drivers on fixture g15's sequence and settings, several seeds: ATE / max error beside the reference loop's"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from test_gpu_slam import _g15_slam
import unislam_amd as us
g = dict(np.load(os.path.join(R, "tests", "golden", "g15_sequence.npz")))
print("reference loop: ATE %.2f cm, max %.2f cm" % (100 * float(g["ate_rmse_m"]), 100 * float(g["err_m"].max())))
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    for prec in ("fp32", "bf16"):
        slam, frames = _g15_slam(us, g, seed=seed, prec=prec)
        slam.run()
        err = (slam.estimate_c2w_list[:, :3, 3] - slam.gt_c2w_list[:, :3, 3]).norm(dim=-1)
        print(f"seed {seed} {prec}: ATE {100 * slam.ate_rmse():.2f} cm, max {100 * float(err.max()):.2f} cm at {int(err.argmax())}, kf {len(slam.mapper.keyframe_list)}, kinds {sorted(slam.mapper._wins)}")
