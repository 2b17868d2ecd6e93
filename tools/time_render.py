"""development: python tools/time_render.py -- the render-only call (MapStep.forward(backward_follows=False)) at the bench shape (4096 rays x 64
samples, room0 tables, 2 x 32 bf16 decoders): one fused launch against four launches, eager and as a replayed graph; ms per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unislam_amd as us
from unislam_amd.graph import CapturedIteration
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import unislam_oracle as O

dev = "cuda:0"
bound = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])          # room0
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": O.per_level_scale(res)}).to(dev)
torch.manual_seed(0)
cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
W = {"fs": 5.0, "center": 200.0, "tail": 10.0, "color": 5.0, "depth": 1.0}
LR = {"decoders": 1e-3, "sdf_grid": 1e-2, "color_grid": 1e-2}
R = 4096
step = us.MapStep(mk(16), mk(19), dec, bound, 48, 16, 0.06, W, LR, max_rays=R)
g = torch.Generator(device=dev).manual_seed(1)
lo, hi = bound[:, 0].to(dev), bound[:, 1].to(dev)
ro = lo + (hi - lo) * (0.3 + 0.4 * torch.rand((R, 3), device=dev, generator=g))
rd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev, generator=g), dim=1)
gd = 1.0 + 2.0 * torch.rand(R, device=dev, generator=g)
gc = torch.rand((R, 3), device=dev, generator=g)
render = lambda: step.forward(ro, rd, gd, gc, has_zero_depth=False, backward_follows=False)

def timed(fn, k=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / k

out = {}
for name, fused, joint in (("one launch", True, False), ("two launches on one stream", False, True), ("four launches on two streams", False, False)):
    step.fused_render, step.render_joint = fused, joint
    out[name] = (round(timed(render), 4), round(timed(CapturedIteration(render, warmup=2).replay), 4))
print("render-only ms (eager, graph replay):", out)
