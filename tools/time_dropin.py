"""time the reference-shaped iteration (bench.py drop_in_bench) by itself, with a host / device split:
   python tools/time_dropin.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench.torch = torch
import unislam_amd as us

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda:0"
bound = bench.load_bound(bench.ROOM0_BOUND)
pls = bench.per_level_scale(816)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": pls}).to(dev)
for prec, hidden in (("bf16", 32), ("fp32", 16)):
    out = bench.drop_in_bench(us, dev, prec, hidden, bound, mk, steps, 10, 4096, 48, 16)
    print(prec, json.dumps({k: (v if not isinstance(v, dict) else {a: round(b, 4) if isinstance(b, float) else b for a, b in v.items()}) for k, v in out.items() if k not in ("workload", "note")}))
