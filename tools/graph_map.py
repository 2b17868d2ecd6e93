"""tools/graph_map.py -- does a hipGraph replay of the mapping iteration beat eager launches?  (timing experiment only: the captured
Adam step count is frozen, so the replayed updates are not the real ones)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import unislam_amd as us
from unislam_amd.graph import CapturedIteration

dev = "cuda:0"
bound = B.load_bound(B.ROOM0_BOUND)
res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
pls = B.per_level_scale(res)
torch.manual_seed(0)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": pls}).to(dev)
cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "fp32"}}
dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
es, ec = mk(16), mk(19)
for overlap in (True, False):
    step = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, B.W, B.LR, max_rays=4096, overlap=overlap)
    ro, rd, gd, gc = B.synthetic_rays(4096, bound, 1000, dev)
    fn = lambda: step.iterate(ro, rd, gd, gc, has_zero_depth=False)
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        fn()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
    it = CapturedIteration(fn)
    for _ in range(20):
        it.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        it.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 200
    ins = step.capture(4096)
    for dst, src in zip(ins, (ro, rd, gd, gc)):
        dst.copy_(src)
    for _ in range(20):
        step.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        step.replay()
    torch.cuda.synchronize(); api = (time.perf_counter() - t0) / 200
    # mixed as in bench.py: every 10th step eager
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200):
        if k % 10 == 0:
            fn()
        else:
            step.replay()
    torch.cuda.synchronize(); mixed = (time.perf_counter() - t0) / 200
    print(f"overlap {overlap}: eager {eager * 1e3:.4f} ms  graph {graph * 1e3:.4f} ms  capture()/replay() {api * 1e3:.4f} ms  mixed {mixed * 1e3:.4f} ms", flush=True)
