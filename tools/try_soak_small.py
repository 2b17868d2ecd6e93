"""development: the soak test's settings (tests/test_gpu_slam.py::_build on the closed loop), several seeds"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from test_gpu_slam import _build
import unislam_amd as us
n = int(sys.argv[1]) if len(sys.argv) > 1 else 132
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    for prec in ("fp32", "bf16"):
        slam, frames = _build(us, n, seed=seed, mlp_precision=prec, room=dict(path="loop", tex_freq=4.0), every=4)
        slam.run()
        err = (slam.estimate_c2w_list[:, :3, 3] - slam.gt_c2w_list[:, :3, 3]).norm(dim=-1)
        tb = 0
        print(f"seed {seed} {prec}: ATE {100 * slam.ate_rmse():.2f} cm, max {100 * float(err.max()):.2f} cm at {int(err.argmax())}, last {100 * float(err[-1]):.2f}, kf {len(slam.mapper.keyframe_list)}, LC {slam.mapper.LC_cnt}, kinds {len(slam.mapper._wins)}", flush=True)
