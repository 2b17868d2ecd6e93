"""kernel trace target: 200 fused tracking iterations (TrackStep.iterate_fused) at the Replica shape (development tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
bench.torch = torch
import unislam_amd as us
dev = "cuda:0"
bound = bench.load_bound(bench.ROOM0_BOUND)
pls = bench.per_level_scale(816)
torch.manual_seed(0)
dec = us.Decoders({"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": os.environ.get("PREC", "bf16")}}, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16, "per_level_scale": pls}).to(dev)
es, ec = mk(16), mk(19)
H, W, fx, fy, cx, cy = 680, 1200, 600.0, 600.0, 599.5, 339.5
W5 = dict(fs=10, center=200, tail=50, color=5, depth=1)
trk = us.TrackStep(es, ec, dec, bound, 32, 8, 0.06, W5, max_rays=2000)
depth = torch.rand(H, W, device=dev) * 2 + 0.5
color = torch.rand(H, W, 3, device=dev)
pose = torch.tensor([1.0, 0.0, 0.0, 0.0, 3.0, 1.2, 0.0], device=dev)
trk.begin_frame(pose, color, depth, 2e-3, 1e-3, H, W, fx, fy, cx, cy, 75, 75)
for _ in range(200):
    trk.iterate_fused(2000)
torch.cuda.synchronize()
