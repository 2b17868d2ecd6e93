"""development: when k_jaccum_p's workgroups start and end at the bench shape, and how many items / records each had.  Needs a timing
build of the library:   touch uni-slam_amd/csrc/hashgrid_joint.hip && make -s -j8 -C uni-slam_amd/csrc EXTRA=-DJ_ACC_TIMING
(restore: the same without EXTRA).  Every workgroup leaves (start, end in 100 MHz ticks, items, records) in the last 32 KiB of the
workspace.   ACC_G=<J_ACCP_GROUPS> python tools/acc_balance.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
import unislam_amd as us
import bench as B

dev = "cuda:0"
import torch as _t
B.torch = _t
bound = B.load_bound(B.ROOM0_BOUND)
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": B.per_level_scale(res)}).to(dev)
torch.manual_seed(0)
cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
es, ec = mk(16), mk(19)
W, LR = B.W, B.LR
st = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, W, LR, max_rays=4096)
c2ws, pool_d, pool_c, pool_dirs = B.keyframe_pools(16, bound, 1000, dev)
win = us.MapWindow(st, c2ws, pool_d, pool_c, pool_dirs, 4096 // 16, joint_opt=False, has_zero_depth=False)
for _ in range(5):
    win.iterate()
torch.cuda.synchronize()
G = int(os.environ.get("ACC_G", "1024"))
d = st.ws[-G * 32:].view(torch.int64).reshape(G, 4).cpu().numpy()
t0, t1, items, rec = d[:, 0], d[:, 1], d[:, 2] & 0xFFFFFFFF, (d[:, 3] >> 32) & 0xFFFFFFFF
ok = t1 > t0
print("workgroups that wrote:", int(ok.sum()))
T0 = t0[ok].min()
s, e = (t0[ok] - T0) / 100.0, (t1[ok] - T0) / 100.0          # us (100 MHz)
dur = e - s
print(f"kernel span {e.max():.1f} us; starts: first wave <= {np.percentile(s, 50):.1f} us (median), last start {s.max():.1f} us")
print(f"busy per workgroup: min {dur.min():.1f}  median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f} us")
print(f"records per workgroup: min {rec[ok].min()}  median {int(np.median(rec[ok]))}  p90 {int(np.percentile(rec[ok], 90))}  max {rec[ok].max()}; sum {rec[ok].sum()}")
print(f"items per workgroup: min {items[ok].min()}  median {int(np.median(items[ok]))}  max {items[ok].max()}")
early = s < 5.0
print(f"first round ({int(early.sum())} workgroups): end median {np.median(e[early]):.1f} p90 {np.percentile(e[early], 90):.1f} max {e[early].max():.1f} us")
late = ~early
if late.any():
    print(f"later rounds ({int(late.sum())}): start median {np.median(s[late]):.1f}, end median {np.median(e[late]):.1f} max {e[late].max():.1f} us")
# rate: records per us of busy time
r = rec[ok] / np.maximum(dur, 1e-3)
print(f"records per us and workgroup: median {np.median(r):.0f}  p10 {np.percentile(r, 10):.0f}  p90 {np.percentile(r, 90):.0f}")
hist, edges = np.histogram(e, bins=12)
print("ends histogram (us):", " ".join(f"{edges[i]:.0f}-{edges[i+1]:.0f}:{hist[i]}" for i in range(len(hist))))
for c in range(4):                                             # by placement: the c-th workgroup on its CU (a fresh launch fills the CUs in order)
    m = slice(c * G // 4, (c + 1) * G // 4)
    print(f"workgroups {c * G // 4:4d}..{(c + 1) * G // 4 - 1:4d}: end mean {e[m].mean():5.1f} us (min {e[m].min():5.1f}, max {e[m].max():5.1f}), items {items[m].mean():.1f}, records {rec[m].mean():.0f}")
