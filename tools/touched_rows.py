"""How many table entries does one mapping iteration touch?  (is a touched-rows Adam worth building?)
Bench workload: 4096 rays x 64 samples from 16 keyframe pools, room0 tables.  Prints, per table and level, the share of entries
whose gradient is non-zero after one iteration, and the share touched at least once over a 15-iteration mapping window."""
import sys, os
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
import bench as B
B.torch = torch
import unislam_amd as us
dev = "cuda:0"
bound = B.load_bound(B.ROOM0_BOUND)
pls = B.per_level_scale(816)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": pls}).to(dev)
torch.manual_seed(0)
dec = us.Decoders({"grid_mode": "hash_grid", "grid": {"tcnn_network": False}}, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
es, ec = mk(16), mk(19)
with torch.no_grad():
    es.params.normal_(0, 0.1); ec.params.normal_(0, 0.1)
step = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, B.W, B.LR, max_rays=4096)
c2ws, pd, pc, pdirs = B.keyframe_pools(B.N_KEYFRAMES, bound, 1000, dev)
P, n_per = pd.shape[1], 4096 // B.N_KEYFRAMES
seen = {"sdf": torch.zeros(es.desc.n_params // 2, dtype=torch.bool, device=dev), "colour": torch.zeros(ec.desc.n_params // 2, dtype=torch.bool, device=dev)}
for it in range(15):
    idx = torch.randint(P, (B.N_KEYFRAMES, n_per), device=dev)
    ro, rd, gd, gc = us.common.get_samples_all(0, 680, 0, 1200, n_per, 680, 1200, 600., 600., 599.5, 339.5, c2ws, pd, pc, dev, pdirs, indices=idx)
    step.forward_backward(ro, rd, gd, gc, has_zero_depth=False)
    for name, enc, o in (("sdf", es, step.o_tab_s), ("colour", ec, step.o_tab_c)):
        g = step.grad[o:o + enc.desc.n_params].view(-1, 2)
        hit = (g != 0).any(dim=1)
        seen[name] |= hit
        if it in (0, 14):
            off = list(enc.desc.offset[:17])
            per = [float(hit[off[l]:off[l + 1]].float().mean()) for l in range(16)]
            cum = [float(seen[name][off[l]:off[l + 1]].float().mean()) for l in range(16)]
            print(f"iteration {it + 1:2d} {name:6s}: touched this iteration {float(hit.float().mean()):.3f} of {hit.numel()} entries; per level",
                  " ".join(f"{p:.2f}" for p in per))
            print(f"             {name:6s}: touched so far {float(seen[name].float().mean()):.3f}; per level", " ".join(f"{p:.2f}" for p in cum))
    step.adam_step()
