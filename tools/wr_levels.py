"""development: where k_jwrite's workgroups spend their time, level by level, at the bench shape.  Needs a timing build of the library:
   touch uni-slam_amd/csrc/hashgrid_joint.hip && make -s -j8 -C uni-slam_amd/csrc EXTRA=-DJ_WR_TIMING      (restore: the same without EXTRA)
Thread 0 of every workgroup leaves a 100 MHz clock at its start, at the start of every level and at its end (24 words per workgroup,
393 216 bytes before the end of the workspace).   python tools/wr_levels.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
import unislam_amd as us
import bench as B
import torch as _t
B.torch = _t
dev = "cuda:0"
bound = B.load_bound(B.ROOM0_BOUND)
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": B.per_level_scale(res)}).to(dev)
torch.manual_seed(0)
cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
es, ec = mk(16), mk(19)
st = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, B.W, B.LR, max_rays=4096)
c2ws, pool_d, pool_c, pool_dirs = B.keyframe_pools(16, bound, 1000, dev)
win = us.MapWindow(st, c2ws, pool_d, pool_c, pool_dirs, 4096 // 16, joint_opt=False, has_zero_depth=False)
for _ in range(5):
    win.iterate()
torch.cuda.synchronize()
n_rows = 4096 * 64 // 512
end = st.ws.numel() & ~7
d = st.ws[end - 393216:end].view(torch.int64).reshape(-1, 24)[:n_rows].cpu().numpy()
T0 = d[:, 0].min()
t = (d[:, :18] - T0) / 100.0                                   # us
print(f"workgroups {n_rows}; starts {t[:, 0].min():.1f} .. {t[:, 0].max():.1f} us; ends {t[:, 17].min():.1f} .. {t[:, 17].max():.1f} us (median {np.median(t[:, 17]):.1f})")
print(f"before level 0 (positions, first cursor set-up): median {np.median(t[:, 1] - t[:, 0]):.2f} us")
lv = np.diff(t[:, 1:18], axis=1)
for l in range(16):
    print(f"  level {l:2d}: median {np.median(lv[:, l]):5.2f}  p10 {np.percentile(lv[:, l], 10):5.2f}  p90 {np.percentile(lv[:, l], 90):5.2f} us")
print(f"sum of the level medians {np.median(lv, axis=0).sum():.1f} us")
for name, m in (("workgroups   0..255 (first on their CU)", slice(0, 256)), ("workgroups 256..511 (second)", slice(256, 512))):
    print(f"{name}: end mean {t[m, 17].mean():.1f} us (min {t[m, 17].min():.1f}, max {t[m, 17].max():.1f})")
print("end by XCD (workgroup index mod 8):", " ".join(f"{t[x::8, 17].mean():.1f}" for x in range(8)))
