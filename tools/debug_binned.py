import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unislam_amd as us
from unislam_amd import _lib as L
DEV = "cuda:0"
pls = 1.2996847159335432
for log2T, n in [(16, 20000)]:
    enc = us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2T,
                                  "base_resolution": 16, "per_level_scale": pls}).to(DEV)
    d = enc.desc
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand((n, 3), device=DEV, generator=g)
    dy = torch.randn((n, 32), device=DEV, generator=g)
    ref = torch.zeros(d.n_params, device=DEV)
    L.check(L.lib().us_hashgrid_bwd_params(ctypes.byref(d), L.ptr(x), L.ptr(dy), n, L.ptr(ref), 0, 0, L.stream()), "ref")
    nbytes = int(L.lib().us_hashgrid_bwd_workspace_bytes(ctypes.byref(d), n))
    offs = np.array(d.offset[:17]) * 2
    for rep in range(4):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        if rep % 2: ws.fill_(0xAB)
        out = torch.zeros(d.n_params, device=DEV)
        L.check(L.lib().us_hashgrid_bwd_binned(ctypes.byref(d), L.ptr(x), L.ptr(dy), n, L.ptr(out), 0, L.ptr(ws), nbytes, L.stream()), "binned")
        torch.cuda.synchronize()
        bad = (~torch.isclose(out, ref, rtol=1e-3, atol=1e-4 * ref.abs().max().item())).nonzero().flatten().cpu().numpy()
        per_level = np.histogram(bad, bins=offs)[0]
        hdr = ws[:3 * (2048 + 64) * 4].view(torch.int32).cpu().numpy()
        counts, offsets = hdr[:2112], hdr[2112:4224]
        if rep == 0 and len(bad):
            import unislam_amd.hashgrid as HG
            idx = HG.grid_indices(d, x).cpu().numpy()          # [n, 16, 8]
            o, r = out.cpu().numpy(), ref.cpu().numpy()
            for bi in bad[:6]:
                lvl = np.searchsorted(offs, bi, side="right") - 1
                ent = (bi - offs[lvl]) // 2
                pts = np.argwhere(idx[:, lvl, :] == ent)
                print("  bad flat", bi, "level", lvl, "entry", ent, "out", o[bi], "ref", r[bi], "diff", o[bi] - r[bi], "contributors (point,corner):", pts[:8].tolist())
        print(f"log2T {log2T} n {n} rep {rep}: mismatches {len(bad)} per level {per_level.tolist()} total_records {offsets[counts.nonzero()[0].max()+1] if counts.any() else 0} sum_counts {counts.sum()}")
