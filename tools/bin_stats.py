"""Per-level bin-load statistics of the binned table-gradient pass on the bench workload (development tool)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
import unislam_amd as us
from unislam_amd import _lib as L
dev = "cuda:0"
bound = B.load_bound(B.ROOM0_BOUND)
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
pls = B.per_level_scale(res)
R = 4096
ro, rd, gd, gc = B.synthetic_rays(R, bound, 1000, dev)
from unislam_amd.renderer import sample_z
z = sample_z(gd, 0.06, torch.linspace(0, 1, 48).to(dev), torch.linspace(0, 1, 16).to(dev), torch.rand(R, 64, device=dev))
pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
x = ((pts.reshape(-1, 3) - bound[:, 0].to(dev)) / (bound[:, 1] - bound[:, 0]).to(dev)).clamp(0, 1).contiguous()
n = x.shape[0]
for l2 in (16, 19):
    enc = us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                  "base_resolution": 16, "per_level_scale": pls}).to(dev)
    d = enc.desc
    dy = torch.randn(n, 32, device=dev)
    g = torch.zeros(d.n_params, device=dev)
    wsb = int(L.lib().us_hashgrid_bwd_workspace_bytes(ctypes.byref(d), n))
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    L.check(L.lib().us_hashgrid_bwd_binned(ctypes.byref(d), L.ptr(x), L.ptr(dy), n, L.ptr(g), 0, L.ptr(ws), wsb, L.stream()), "b")
    torch.cuda.synchronize()
    hdr = ws[:2 * (4096 + 64) * 4].view(torch.int32).cpu().numpy().astype(np.int64)
    tot = hdr[:4096]
    # bins per level, as make_binmap
    want = 0
    while (8192 << want) < n * 8 and want < 8: want += 1
    first = 0
    print(f"log2T {l2}: records {tot.sum()} of {n*128}  max bin {tot.max()}  mean {tot[tot>0].mean():.0f}")
    for l in range(16):
        hs = d.offset[l + 1] - d.offset[l]
        lg = 0
        while (2048 << lg) < hs: lg += 1
        lg = max(lg, want)
        while lg > 0 and (1 << lg) > hs: lg -= 1
        nb = 1 << lg
        t = tot[first:first + nb]
        print(f"  level {l:2d} res {d.resolution[l]:4d} entries {hs:7d} bins {nb:4d}  records {t.sum():8d} ({t.sum()/(n*8):.2f} of max)  bin max {t.max():6d} mean {t.mean():8.0f}  max/mean {t.max()/max(t.mean(),1):.2f}")
        first += nb
