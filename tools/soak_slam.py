"""development: a long run of the SLAM drivers on the synthetic room with the per-frame pose error, the activated-mapping state and the
number of keyframes printed: python tools/soak_slam.py n_frames [graph_replay 0|1] [cfg overrides dict] [SyntheticRoom kwargs dict]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch
import unislam_amd as us
import unislam_oracle as O
from unislam_amd.synthetic import SyntheticRoom
from unislam_amd.slam import SLAM
DEV = "cuda:0"
n = int(sys.argv[1]); replay = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
torch.manual_seed(0)
room_kw = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
frames = SyntheticRoom(n_frames=n, H=680, W=1200, device=DEV, **room_kw)
bound = O.load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]])
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16,
                   "per_level_scale": O.per_level_scale(res)}
es, ec = us.HashGridEncoding(3, ecfg(16)).to(DEV), us.HashGridEncoding(3, ecfg(19)).to(DEV)
cfg = {"rendering": {"perturb": True, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid",
       "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
dec.bound = bound
extra_cfg = {}
if len(sys.argv) > 3: extra_cfg = eval(sys.argv[3])
slam = SLAM(frames, (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy), es, ec, dec, bound,
            cfg={"mapping": dict(iters_first=100, graph_replay=replay, **extra_cfg.get("mapping", {})), "tracking": dict(graph_replay=replay, **extra_cfg.get("tracking", {}))})
def log(idx, s):
    e = float((s.estimate_c2w_list[idx][:3, 3] - s.gt_c2w_list[idx][:3, 3]).norm()) * 100
    w = s.tracker.rendered_weight.get(idx)
    d = (s.estimate_c2w_list[idx][:3, 3] - s.gt_c2w_list[idx][:3, 3]) * 100
    print(f"{idx:4d} err {e:7.3f} cm ({float(d[0]):6.2f} {float(d[1]):6.2f} {float(d[2]):6.2f})  tb {int(s.tracking_back)}  kf {len(s.mapper.keyframe_list):3d}  unc {float(w) if w is not None else -1:.5f} joint {int(s.mapper.joint_opt)}", flush=True)
slam.run(log=log)
from unislam_amd import graph
print("ATE", 100 * slam.ate_rmse(), "max mem GB", torch.cuda.max_memory_allocated() / 2**30, "captured graphs alive", len(graph._KEEP))
