"""host-side profile of the reference-shaped iteration: cProfile over 100 iterations + GPU-busy time from the kernel sum
   python tools/prof_dropin.py"""
import cProfile, pstats, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench.torch = torch
import unislam_amd as us

dev = "cuda:0"
bound = bench.load_bound(bench.ROOM0_BOUND)
pls = bench.per_level_scale(816)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": pls}).to(dev)
CAM, W, LR = bench.CAM, bench.W, bench.LR
c2ws, pd, pc, pr = bench.keyframe_pools(16, bound, 5000, dev)
cfg = {"rendering": {"perturb": True, "n_stratified": 48, "n_importance": 16}, "scale": 1, "grid_mode": "hash_grid",
       "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
rend = us.Renderer(cfg, types.SimpleNamespace(bound=bound, device=dev, H=CAM["H"], W=CAM["W"], fx=CAM["fx"], fy=CAM["fy"], cx=CAM["cx"], cy=CAM["cy"]))
torch.manual_seed(0)
dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
es, ec = mk(16), mk(19)
opt = us.optim.Adam([{"params": list(dec.parameters()), "lr": 1e-3}, {"params": [es.params], "lr": 0.05}, {"params": [ec.params], "lr": 0.05}])

def it():
    opt.zero_grad()
    ro, rd, gd, gc = us.common.get_samples_all(0, 680, 0, 1200, 256, 680, 1200, 600., 600., 599.5, 339.5, c2ws, pd, pc, dev, pr)
    inside = us.common.bbox_filter(ro, rd, gd, bound)
    ret = rend.render_batch_ray(([es], [ec]), dec, rd, ro, dev, 0.06, gt_depth=gd)
    loss = us.mapping_loss(ret, gd, gc, 0.06, W, valid=inside)
    loss.backward()
    opt.step()

for _ in range(20):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    it()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time {10 * t_host:.3f} ms/iter, wall {10 * t_all:.3f} ms/iter")
pr_ = cProfile.Profile()
pr_.enable()
for _ in range(100):
    it()
pr_.disable()
torch.cuda.synchronize()
pstats.Stats(pr_).sort_stats("tottime").print_stats(28)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
    for _ in range(20):
        it()
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
