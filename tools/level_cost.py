"""development: where the encoder's time goes, level by level -- a one-level grid per level of the room0 tables (same resolution to
rounding, same table size), encoded for the bench's 4096 x 64 sample points (rays from the keyframe pools); us per launch, the share of
lanes whose cell equals the previous lane's (= what a "run heads only" gather would skip) and the distinct 128-byte lines per wave."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
bench.torch = torch
import unislam_amd as us
from unislam_amd import _lib as L
dev = "cuda:0"
bound = bench.load_bound(bench.ROOM0_BOUND)
pls = bench.per_level_scale(816)
torch.manual_seed(0)
dec = us.Decoders({"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16, "per_level_scale": pls}).to(dev)
es, ec = mk(16), mk(19)
step = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, bench.W, bench.LR, max_rays=4096)
c2ws, pd, pc, pr = bench.keyframe_pools(16, bound, 1000, dev)
win = us.MapWindow(step, c2ws, pd, pc, pr, 256, joint_opt=False, has_zero_depth=False)
win.iterate()
x = step.pts[:4096].reshape(-1, 3).clamp(0, 1).contiguous()
n = x.shape[0]
lib, st = L.lib(), L.stream()
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tot = {16: 0.0, 19: 0.0}
for l in range(16):
    res = int(es.desc.resolution[l])
    scale = float(es.desc.scale[l])
    cell = torch.floor(x * scale + 0.5).long()
    key = (cell[:, 0] * 1024 + cell[:, 1]) * 1024 + cell[:, 2]
    k2 = key.reshape(-1, 64)
    dup = float((k2[:, 1:] == k2[:, :-1]).float().mean())
    row = [f"level {l:2d} res {res:4d}: same cell as the previous sample {100 * dup:5.1f} %"]
    for l2 in (16, 19):
        g = us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 1, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": res, "per_level_scale": 1.0}).to(dev)
        out = torch.empty(n, 2, device=dev)
        us_ = t(lambda: L.check(lib.us_hashgrid_fwd(ctypes.byref(g.desc), L.ptr(g.params.detach()), L.ptr(x), n, L.ptr(out), None, 3, st), "f"))
        tot[l2] += us_
        row.append(f"log2T {l2}: {int(g.desc.offset[1]) * 8 / 1024:8.0f} KiB {us_:6.1f} us")
    print("   ".join(row))
print(f"sum over the levels as separate launches: sdf {tot[16]:.1f} us, colour {tot[19]:.1f} us (each launch pays ~4.5 us of its own)")
