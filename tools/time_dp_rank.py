"""development: python tools/time_dp_rank.py -- what ONE rank of the data-parallel step costs (4096 rays x 64, room0 tables, bf16 decoders): a
1-rank RCCL group, so the collectives are issued and waited for but move nothing; MapStep(group=True) in its two
dp_modes, and the single-process step (eager) beside them.  ms per step."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "oracle"))
import unislam_amd as us
from unislam_amd.dist import dp_iterate
import unislam_oracle as O

dist.init_process_group("nccl", rank=0, world_size=1)
dev = "cuda:0"
bound = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": O.per_level_scale(res)}).to(dev)
W = {"fs": 5.0, "center": 200.0, "tail": 10.0, "color": 5.0, "depth": 1.0}
LR = {"decoders": 1e-3, "sdf_grid": 1e-2, "color_grid": 1e-2}
R = 4096
g = torch.Generator(device=dev).manual_seed(1)
lo, hi = bound[:, 0].to(dev), bound[:, 1].to(dev)
ro = lo + (hi - lo) * (0.3 + 0.4 * torch.rand((R, 3), device=dev, generator=g))
rd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev, generator=g), dim=1)
gd = 1.0 + 2.0 * torch.rand(R, device=dev, generator=g)
gc = torch.rand((R, 3), device=dev, generator=g)

def build(**kw):
    torch.manual_seed(0)
    cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
    dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
    return us.MapStep(mk(16), mk(19), dec, bound, 48, 16, 0.06, W, LR, max_rays=R, **kw)

def timed(fn, k=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return round(1e3 * (time.perf_counter() - t) / k, 4)

out = {}
for name, kw in (("dp local_fast (default: joint kernels, side streams, accumulate split per grid)", dict(group=True)),
                 ("dp colour_first (one-grid kernels, one stream)", dict(group=True, dp_mode="colour_first")),
                 ("dp local_fast, one stream", dict(group=True, overlap=False)),
                 ("dp one-grid + two streams", dict(group=True, dp_mode="colour_first", overlap=True))):
    st = build(**kw)
    out[name] = timed(lambda: dp_iterate(st, (ro, rd, gd, gc, None, False), group=True))
st = build()
out["single process, eager"] = timed(lambda: st.iterate(ro, rd, gd, gc, has_zero_depth=False))
print(out)
dist.destroy_process_group()
