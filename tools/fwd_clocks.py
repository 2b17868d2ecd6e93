"""development: when the encoder's workgroups (k_jfwd, counting form: 1024 points x one level) start, have their features out and end,
at the bench shape.  Needs a timing build:  touch uni-slam_amd/csrc/hashgrid_joint.hip && make -s -j8 -C uni-slam_amd/csrc EXTRA=-DJ_FWD_TIMING
(restore: the same without EXTRA).  python tools/fwd_clocks.py"""
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
nx = 4096 * 64 // 1024
end = st.ws.numel() & ~7
d = st.ws[end - 393216:end].view(torch.int64).reshape(-1, 4)[:16 * nx].cpu().numpy().reshape(16, nx, 4)
T0 = d[..., 0].min()
s, f, e = (d[..., 0] - T0) / 100.0, (d[..., 1] - T0) / 100.0, (d[..., 2] - T0) / 100.0
print(f"workgroups {16 * nx}; kernel span {e.max():.1f} us")
for l in range(16):
    print(f"  level {l:2d}: starts {s[l].min():6.1f} .. {s[l].max():6.1f}   ends {e[l].min():6.1f} .. {e[l].max():6.1f}   workgroup: median {np.median(e[l] - s[l]):5.1f} us, "
          f"features out after {np.median(f[l] - s[l]):5.1f}, counting {np.median(e[l] - f[l]):4.1f}")
dur = e - s
xcc = (d[..., 3] >> 32) & 0xF
print("levels by XCC id of their workgroups:", " ".join(f"{l}:{sorted(set(xcc[l].tolist()))}" for l in range(16)))
for x in sorted(set(xcc.ravel().tolist())):
    m = xcc == x
    print(f"  XCC {x}: workgroups {int(m.sum())}, last end {e[m].max():.1f} us")
print(f"workgroup time: median {np.median(dur):.1f} us; sum / 512 resident = {dur.sum() / 512:.1f} us")
# how many workgroups run at a time
ev = sorted([(x, 1) for x in s.ravel()] + [(x, -1) for x in e.ravel()])
run, t_prev, area = 0, 0.0, 0.0
for t, k in ev:
    area += run * (t - t_prev); t_prev = t; run += k
print(f"workgroups in flight, time average: {area / e.max():.0f}")
