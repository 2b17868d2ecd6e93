"""BASELINE configs[2] shape (ScanNet scene0000 tables, 8192 rays x 96 samples, 25 % zero-depth rays): ms per mapping iteration
with the zero-depth branch on, and without zero-depth rays (development tool)."""
import sys, os, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "oracle"))
import torch
import unislam_amd as us
import unislam_oracle as O
DEV = "cuda:0"
W = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
torch.manual_seed(0)
bound = O.load_bound([[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]])
ecfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 16, "base_resolution": 16,
        "per_level_scale": O.per_level_scale(456)}
cfg = {"rendering": {"perturb": True, "n_stratified": 80, "n_importance": 16}, "scale": 1, "grid_mode": "hash_grid", "grid": {"tcnn_network": False}}
R = 8192
g = torch.Generator().manual_seed(3)
ro = bound.mean(1)[None].repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05
rd = torch.randn(R, 3, generator=g); rd = rd / rd.norm(dim=-1, keepdim=True)
far = O.bbox_far(ro, rd, bound)
for frac_zero in (0.25, 0.0):
    es, ec = us.HashGridEncoding(3, ecfg).to(DEV), us.HashGridEncoding(3, ecfg).to(DEV)
    with torch.no_grad():
        es.params.mul_(2000); ec.params.mul_(2000)
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06).to(DEV)
    gd = torch.minimum(torch.rand(R, generator=g) * 3 + 0.5, 0.9 * far)
    if frac_zero > 0:
        gd[::4] = 0.0
    gc = torch.rand(R, 3, generator=g)
    step = us.MapStep(es, ec, dec, bound, 80, 16, 0.06, W, dict(decoders=0.001, sdf_grid=0.02, color_grid=0.02), max_rays=R)
    a = [t.to(DEV) for t in (ro, rd, gd, gc)]
    hz = frac_zero > 0
    for _ in range(5):
        step.iterate(*a, has_zero_depth=hz)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        step.iterate(*a, has_zero_depth=hz)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"cfg3 8192 x 96, zero-depth fraction {frac_zero}: {ms:.3f} ms per iteration = {R / ms * 1e3 / 1e6:.2f} M rays/s")
