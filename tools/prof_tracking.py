import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import unislam_amd as us
dev = "cuda:0"
bound = bench.load_bound(bench.ROOM0_BOUND)
pls = bench.per_level_scale(816)
torch.manual_seed(0)
dec = us.Decoders({"grid_mode": "hash_grid", "grid": {"tcnn_network": False}}, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16, "per_level_scale": pls}).to(dev)
es, ec = mk(16), mk(19)
print(bench.tracking_bench(us, es, ec, dec, bound, dev, iters=100))
