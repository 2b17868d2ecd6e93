"""development: wall time per frame of the tracking / mapping drivers (unislam_amd.slam) at Replica's settings -- 680 x 1200 frames,
2000 x 8 tracking, 4000 x 15 mapping every 4th frame, tables 2^16 / 2^19 at 1 cm -- on the synthetic room (frames rendered ahead).
   python tools/time_slam.py [n_frames] [graph_replay 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import unislam_amd as us
import unislam_oracle as O
from unislam_amd.synthetic import SyntheticRoom
from unislam_amd.slam import SLAM

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
replay = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
torch.manual_seed(0)
frames = SyntheticRoom(n_frames=n, H=680, W=1200, device=DEV)
for i in range(n):
    frames[i]
bound = O.load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]])
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16,
                   "per_level_scale": O.per_level_scale(res)}
es, ec = us.HashGridEncoding(3, ecfg(16)).to(DEV), us.HashGridEncoding(3, ecfg(19)).to(DEV)
cfg = {"rendering": {"perturb": True, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid",
       "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
dec.bound = bound
slam = SLAM(frames, (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy), es, ec, dec, bound,
            cfg={"mapping": dict(iters_first=100, graph_replay=replay)})
t_track, t_map, marks = [], [], []
tr, mp = slam.tracker.track_frame, slam.mapper.map_frame


def timed(fn, sink):
    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); sink.append(time.perf_counter() - t0)
        return r
    return f


slam.mapper.timing = {} if os.environ.get("US_SLAM_TIMING") else None
slam.tracker.track_frame = timed(tr, t_track)
slam.mapper.map_frame = timed(mp, t_map)
torch.cuda.synchronize(); t0 = time.perf_counter()
slam.run()
torch.cuda.synchronize(); el = time.perf_counter() - t0
med = lambda xs: sorted(xs)[len(xs) // 2] if xs else float("nan")
print(f"frames {n} graph_replay {replay}: {1e3 * el / n:.2f} ms per frame ({n / el:.1f} frames/s); tracking median {1e3 * med(t_track):.2f} ms "
      f"(min {1e3 * min(t_track):.2f}), mapped frames {len(t_map)}: median {1e3 * med(t_map[1:]):.2f} ms (first {1e3 * t_map[0]:.1f}); "
      f"ATE {100 * slam.ate_rmse():.2f} cm; keyframes {len(slam.mapper.keyframe_list)}")
print("tracking ms:", " ".join(f"{1e3 * t:.1f}" for t in t_track[:24]))
print("mapping  ms:", " ".join(f"{1e3 * t:.1f}" for t in t_map[:24]))
if slam.mapper.timing:
    for k, v in slam.mapper.timing.items():
        print(f"  {k:36s} median {sorted(v)[len(v) // 2]:.3f} ms   all: " + " ".join(f"{x:.2f}" for x in v[:14]))
