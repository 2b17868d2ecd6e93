"""
GPU: the mapping window with joint pose optimisation (unislam_amd.window.MapWindow; the reference's default joint_opt,
src/Mapper.py:359-376,443-459) -- its three kernels against torch / the oracle, the whole iteration against the fixture generated from
the reference's Mapper.optimize_mapping (g14), hipGraph replay against eager.
"""
import copy
import os

import numpy as np
import pytest
import torch

import unislam_oracle as O
from test_oracle_golden import g14_window

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BOUND = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])
W = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
LR = dict(decoders=0.001, sdf_grid=0.05, color_grid=0.05)
T = torch.from_numpy


def _cfg(tcnn=False, ns=32, ni=8):
    return {"rendering": {"perturb": True, "n_stratified": ns, "n_importance": ni}, "scale": 1, "grid_mode": "hash_grid",
            "grid": {"tcnn_network": tcnn}}


def _ecfg(log2T, res=816):
    return {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2T,
            "base_resolution": 16, "per_level_scale": O.per_level_scale(res)}


def _window(b, P, seed, spread=0.1):
    """a synthetic window: poses around a base pose inside the room, pools with depths inside the box (a few beyond it)"""
    g = torch.Generator().manual_seed(seed)
    base = torch.tensor([0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0])
    poses = base[None] + torch.cat([torch.randn(b, 4, generator=g) * 0.05, torch.randn(b, 3, generator=g) * spread], -1)
    c2ws = O.cam_pose_to_matrix(poses)
    u, v = torch.rand(b, P, generator=g) * 31, torch.rand(b, P, generator=g) * 23
    dirs = torch.stack([(u - 15.5) / 18.0, -(v - 11.5) / 18.0, -torch.ones_like(u)], -1)
    depths = torch.rand(b, P, generator=g) * 1.5 + 0.5
    depths[:, 5::41] = 50.0
    colors = torch.rand(b, P, 3, generator=g)
    return c2ws, depths, colors, dirs


def test_window_rays_and_pose_step_against_autograd():
    """us_window_rays == the reference's cam_pose_to_matrix + get_samples_all (both calls); us_pose_window_step's gradient == autograd
    through them, per frame, with and without the extra block; its Adam step == torch.optim.Adam on the poses"""
    import ctypes
    import unislam_amd as us
    from unislam_amd import _lib as L
    lib, P_ = L.lib(), L.ptr
    for b, n_per, extra in ((6, 20, None), (12, 10, (10, 37)), (4, 9, (10, 5)), (2, 33, None)):
        c2ws, depths, colors, dirs = _window(b, 300, 50 + b)
        g = torch.Generator().manual_seed(b)
        idx = torch.randint(300, (b, n_per), generator=g)
        nf = min(extra[0], b) if extra else 0
        idx2 = torch.randint(300, (nf, extra[1]), generator=g) if extra else None
        poses = O.matrix_to_cam_pose(c2ws[1:]).requires_grad_(True)
        ro, rd, gd, gc = O.window_rays(c2ws[0], poses, depths, colors, dirs, idx, (nf, extra[1]) if extra else None, idx2)
        R = ro.shape[0]
        f = lambda *s: torch.empty(s, device=DEV)
        o_ro, o_rd, o_gd, o_gc, o_dirs = f(R, 3), f(R, 3), f(R), f(R, 3), f(R, 3)
        d_c0, d_p = c2ws[0].contiguous().to(DEV), poses.detach().contiguous().to(DEV)
        pd, pc, pr = depths.to(DEV), colors.to(DEV), dirs.contiguous().to(DEV)
        idx_dev, idx2_dev = idx.to(DEV), (idx2.to(DEV) if extra else None)
        L.check(lib.us_window_rays(P_(d_c0), P_(d_p), P_(pd), P_(pc), P_(pr), P_(idx_dev), 300, 0, b, n_per, P_(o_ro), P_(o_rd), P_(o_gd),
                                   P_(o_gc), P_(o_dirs), L.stream()), "us_window_rays")
        if extra:
            r0 = b * n_per
            off = lambda t, w: ctypes.c_void_p(t.data_ptr() + 4 * r0 * w)
            L.check(lib.us_window_rays(P_(d_c0), P_(d_p), P_(pd), P_(pc), P_(pr), P_(idx2_dev), 300, b - nf, nf, extra[1], off(o_ro, 3),
                                       off(o_rd, 3), off(o_gd, 1), off(o_gc, 3), off(o_dirs, 3), L.stream()), "us_window_rays")
        assert torch.allclose(o_rd.cpu(), rd.detach(), rtol=1e-5, atol=1e-6) and torch.allclose(o_ro.cpu(), ro.detach(), rtol=1e-6, atol=1e-7)
        assert torch.equal(o_gd.cpu(), gd) and torch.equal(o_gc.cpu(), gc)
        g_o, g_d = torch.randn(R, 3, generator=g), torch.randn(R, 3, generator=g)
        torch.autograd.backward([ro, rd], [g_o, g_d])
        g7, go_dev, gd_dev = f(b - 1, 7), g_o.to(DEV), g_d.to(DEV)
        args = (b - 1, P_(go_dev), P_(gd_dev), P_(o_dirs), n_per, n_per, max(b - nf - 1, 0),
                b * n_per + (extra[1] if (extra and nf == b) else 0), extra[1] if extra else 0)
        L.check(lib.us_pose_window_step(P_(d_p), *args, None, None, P_(g7), 0.0, 0.0, 0.9, 0.999, 1e-8, None, L.US_POSE_GRAD_ONLY, L.stream()),
                "us_pose_window_step")
        assert torch.allclose(g7.cpu(), poses.grad, rtol=1e-4, atol=1e-5 * float(poses.grad.abs().max())), (b, extra)
        # the optimiser step: three steps of torch.optim.Adam, each on the gradient of <g_o, rays_o> + <g_d, rays_d> at the moved poses
        # (what the kernel re-derives from the poses it reads), the step count advanced by us_adam_step_inc as in MapWindow
        pt = torch.nn.Parameter(poses.detach().clone()); opt = torch.optim.Adam([pt], lr=1e-3)
        m7, v7, step_dev = torch.zeros(b - 1, 7, device=DEV), torch.zeros(b - 1, 7, device=DEV), torch.zeros(8, device=DEV)
        for _ in range(3):
            ro_t, rd_t, _, _ = O.window_rays(c2ws[0], pt, depths, colors, dirs, idx, (nf, extra[1]) if extra else None, idx2)
            opt.zero_grad(); torch.autograd.backward([ro_t, rd_t], [g_o, g_d]); opt.step()
            L.check(lib.us_adam_step_inc(P_(step_dev), 0.9, 0.999, L.stream()), "us_adam_step_inc")
            L.check(lib.us_pose_window_step(P_(d_p), *args, P_(m7), P_(v7), None, 1e-3, 1e-3, 0.9, 0.999, 1e-8, P_(step_dev), 0, L.stream()),
                    "us_pose_window_step")
        assert torch.allclose(d_p.cpu(), pt.detach(), rtol=0, atol=2e-6), float((d_p.cpu() - pt.detach()).abs().max())


@pytest.mark.parametrize("S,pair", [(64, (16, 19)), (40, (14, 15)), (96, (16, 16)), (7, (12, 12))])
def test_input_gradient_of_both_grids_reduced_to_rays(S, pair):
    """us_hashgrid_bwd_input_rays: per-point dL/dx bit-identical to two us_hashgrid_bwd_input_gather launches, dL/do and dL/dd equal to
    us_ray_points_bwd of it (summation order differs: 1e-5); the stored-derivative path (half-precision dy/dx planes) against both"""
    import ctypes
    import unislam_amd as us
    from unislam_amd import _lib as L
    lib, P_ = L.lib(), L.ptr
    R = 133
    N = R * S
    g = torch.Generator().manual_seed(S)
    es, ec = us.HashGridEncoding(3, _ecfg(pair[0])).to(DEV), us.HashGridEncoding(3, _ecfg(pair[1])).to(DEV)
    with torch.no_grad():
        es.params.copy_(torch.randn(es.params.shape, generator=g) * 0.3); ec.params.copy_(torch.randn(ec.params.shape, generator=g) * 0.3)
    x = (torch.rand(N, 3, generator=g) * 1.1 - 0.05).to(DEV)                   # some coordinates outside [0,1]: the clamp's zero gradient
    dya, dyb = torch.randn(16, N, 2, generator=g).to(DEV), torch.randn(16, N, 2, generator=g).to(DEV)
    z = torch.rand(R, S, generator=g).to(DEV) * 3
    bh = us.common.bound_host(BOUND)
    f = lambda *s: torch.empty(s, device=DEV)
    d_ref, go_ref, gd_ref = f(N, 3), f(R, 3), f(R, 3)
    ds, dc = ctypes.byref(es.desc), ctypes.byref(ec.desc)
    L.check(lib.us_hashgrid_bwd_input_gather(ds, P_(es.params.detach()), P_(x), P_(dya), N, P_(d_ref), 3, L.stream()), "a")
    L.check(lib.us_hashgrid_bwd_input_gather(dc, P_(ec.params.detach()), P_(x), P_(dyb), N, P_(d_ref), 3 | L.US_GRID_ACCUMULATE, L.stream()), "b")
    L.check(lib.us_ray_points_bwd(P_(d_ref), P_(z), bh, R, S, P_(go_ref), P_(gd_ref), L.stream()), "c")
    d_new, go, gd = f(N, 3), f(R, 3), f(R, 3)
    assert lib.us_hashgrid_bwd_input_rays_supported(ds, dc, S) == 1
    L.check(lib.us_hashgrid_bwd_input_rays(ds, dc, P_(es.params.detach()), P_(ec.params.detach()), P_(x), P_(dya), P_(dyb), R, S, P_(z), bh,
                                           P_(go), P_(gd), P_(d_new), 3, L.stream()), "us_hashgrid_bwd_input_rays")
    assert torch.equal(d_new, d_ref)
    for a, b in ((go, go_ref), (gd, gd_ref)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))
    go2, gd2 = f(R, 3), f(R, 3)                                                # without the per-point output
    L.check(lib.us_hashgrid_bwd_input_rays(ds, dc, P_(es.params.detach()), P_(ec.params.detach()), P_(x), P_(dya), P_(dyb), R, S, P_(z), bh,
                                           P_(go2), P_(gd2), None, 3, L.stream()), "us_hashgrid_bwd_input_rays")
    assert torch.equal(go2, go) and torch.equal(gd2, gd)
    assert lib.us_hashgrid_bwd_input_rays_supported(ds, dc, 129) == 0
    # the stored-derivative path: the joint encoder leaves dy/dx (planes [L][3][N][2]), one streaming launch contracts it
    fa, fb, fa0, fb0 = f(16, N, 2), f(16, N, 2), f(16, N, 2), f(16, N, 2)
    dda, ddb = (torch.empty((16, 3, N, 2), dtype=torch.float16, device=DEV) for _ in range(2))      # planes [L][3][N][2] of halves
    L.check(lib.us_hashgrid_fwd_joint(ds, dc, P_(es.params.detach()), P_(ec.params.detach()), P_(x), N, P_(fa0), P_(fb0), 3, None, 0, L.stream()), "fwd")
    L.check(lib.us_hashgrid_fwd_joint_dydx(ds, dc, P_(es.params.detach()), P_(ec.params.detach()), P_(x), N, P_(fa), P_(fb), P_(dda), P_(ddb), 3,
                                           None, 0, L.stream()), "us_hashgrid_fwd_joint_dydx")
    assert torch.equal(fa, fa0) and torch.equal(fb, fb0)
    # dy/dx against the one-grid encoder's stored tensor [N][C][3]
    ref_dd = f(N, 32, 3); tmp = f(N, 32)
    L.check(lib.us_hashgrid_fwd(dc, P_(ec.params.detach()), P_(x), N, P_(tmp), P_(ref_dd), 1, L.stream()), "us_hashgrid_fwd")
    assert torch.equal(ddb.permute(2, 0, 3, 1).reshape(N, 32, 3), ref_dd.half())   # the one-grid encoder's values, rounded once to half
    d3, go3, gd3 = f(N, 3), f(R, 3), f(R, 3)
    L.check(lib.us_hashgrid_dydx_rays(16, P_(dya), P_(dyb), P_(dda), P_(ddb), R, S, P_(z), bh, P_(go3), P_(gd3), P_(d3), L.stream()), "us_hashgrid_dydx_rays")
    # the stored halves carry 2^-11 per value: the 96 products of a point's gradient agree with the re-gathering path to ~1e-3 of its size
    assert float((d3 - d_ref).norm() / d_ref.norm()) < 5e-4
    for a, b in ((go3, go), (gd3, gd)):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-4 * float(b.abs().max()))


def _g14_scene(us, g):
    dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06)
    dec.load_state_dict({k[len("dec0__"):].replace("__", "."): T(v) for k, v in g.items() if k.startswith("dec0__")})
    dec = dec.to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(10, 64)).to(DEV), us.HashGridEncoding(3, _ecfg(10, 64)).to(DEV)
    with torch.no_grad():
        es.params.copy_(T(g["grid_s0"])); ec.params.copy_(T(g["grid_c0"]))
    return dec, es, ec


def _tables_after_adam(got, want, rtol=1e-3, atol=2e-5, max_outliers=1e-3, outlier_atol=2e-3, cancel=None):
    """table entries after Adam steps: rtol / atol for all but a few.  Adam divides an entry's step by that entry's own gradient history, so
    an entry whose gradient is a near-cancelling sum (|g| at the rounding level of its contributions) takes a step whose size -- a visible
    fraction of lr = 0.05 -- depends on the order of a float sum: the reference's CPU sum and the kernels' f64 sum differ there.  At most
    0.1 % of the entries may sit outside the bar (r5, regenerated g14: 16 of 32 768), and those within 2e-3."""
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert bad.mean() <= max_outliers, (int(bad.sum()), bad.size)
    assert float(np.abs(got - want).max()) <= outlier_atol, float(np.abs(got - want).max())
    if cancel is not None and bad.any():
        # ... and the entries outside the bar are of the kind that CAN differ between two correct implementations (an indexing bug
        # must not hide in the allowance): the fixture's gradient is an fp32 sum formed entry by entry on the CPU, the kernels sum in f64.  An
        # fp32 sum of N contributions of total magnitude M is off by up to ~N eps M, i.e. by N eps / ratio relative to a gradient that cancels
        # to ratio = |g| / M of its contributions' magnitude -- percents for the HEAVIEST entries of these 2^10-entry tables (thousands of
        # colliding contributions) once they cancel ten- to hundred-fold, and Adam turns a relative error of g into the same relative error
        # of a step of ~lr.  Measured (r6): all 16 outliers of g14 carry 400 - 2300 x the median touched entry's mass at ratios 0.01 - 0.13.
        # cancel[k] = (ratio, M) of iteration k per entry; score = M / ratio ranks the entries by that error bound.
        ratio = np.min(np.stack([c[0] for c in cancel]), axis=0)
        mass = np.max(np.stack([c[1] for c in cancel]), axis=0)
        touched = mass > 0
        bound = mass / np.median(mass[touched]) * 6e-8 / np.maximum(ratio, 1e-12)   # ~ N eps / ratio with N ~ mass in units of a typical entry's
        others = float(np.median(bound[touched & ~bad]))
        print("outliers: mass / median", np.round(mass[bad] / np.median(mass[touched]), 1), "ratio", np.round(ratio[bad], 3),
              "N eps / ratio", np.array2string(bound[bad], precision=1), "median of the other touched entries", others)
        # (a random walk, not a worst case: no sharp cut separates them, but the outliers come from the upper tail of this bound)
        assert float(np.median(bound[bad])) > 5.0 * others, (float(np.median(bound[bad])), others)


def _cancellation(step, which):
    """per entry of one table: |gradient| / sum of |contributions| of the iteration that has just run (1 = all contributions of one sign,
    ~1e-7 = a sum that cancels to its rounding level).  The denominator is the table gradient of |dL/dfeatures| (positive weights)."""
    import ctypes
    from unislam_amd import _lib as L
    enc, off, dfeat = (step.es, step.o_tab_s, step.d_feat_s) if which == "s" else (step.ec, step.o_tab_c, step.d_feat_c)
    n = step.n_rays * step.S
    g = step.grad[off:off + enc.desc.n_params].detach().clone()
    a = torch.zeros_like(g)
    dy = dfeat[:n * 32].abs().contiguous() if dfeat.dim() == 1 else dfeat.reshape(-1)[:n * 32].abs().contiguous()
    L.check(L.lib().us_hashgrid_bwd_params(ctypes.byref(enc.desc), L.ptr(step.pts), L.ptr(dy), n, L.ptr(a), 0, 3, L.stream()), "abs gradient")
    return (g.abs() / a.clamp_min(1e-30)).cpu().numpy() * (a > 0).cpu().numpy(), a.cpu().numpy()


@pytest.mark.parametrize("tag", ["w6", "w12x"])
def test_mapwindow_reproduces_reference_joint_opt(golden, tag):
    """MapWindow driven like Mapper.optimize_mapping with joint_opt (6-frame window; 12-frame window + the 10 x 200 extra rays) against
    fixture g14: pose / table / beta gradients of the first iteration, and tables, decoders and poses after two optimiser steps"""
    import unislam_amd as us
    g = golden("g14_mapping_joint")
    for iters in (1, 2):
        dec, es, ec = _g14_scene(us, g)
        c2ws, depths, colors, dirs, n_per, extra, frames = g14_window(g, tag)     # re-seeds torch's CPU generator like the reference run
        b, P = depths.shape
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=b * n_per + (2000 if extra else 0))
        step.reset_optimizer(float(g["lr_factor"]))
        win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, cam_lr=float(g["cam_lr"]), extra=extra)
        cancel_s, cancel_c = [], []
        for it in range(iters):
            idx = torch.randint(P, (n_per * b,)).reshape(b, -1)
            idx2 = torch.randint(P, (extra[1] * extra[0],)).reshape(extra[0], -1) if extra else None
            # the reference jitters only the rays its pre-filter kept (src/Mapper.py:396-406 -> Renderer.py:55): same stream positions
            ro, rd, gd, _ = O.window_rays(c2ws[0], win.poses.cpu(), depths, colors, dirs, idx, extra, idx2)
            inside = O.bbox_far(ro, rd, BOUND) >= gd
            t_rand = torch.zeros(ro.shape[0], 40)
            t_rand[inside] = torch.rand(int(inside.sum()), 40)
            win.iterate(idx.to(DEV), idx2.to(DEV) if extra else None, t_rand=t_rand.to(DEV))
            cancel_s.append(_cancellation(step, "s")); cancel_c.append(_cancellation(step, "c"))
            assert torch.equal(step.valid[:win.R].bool().cpu(), inside)
            assert int(step.zd_count) == 0                       # has_zero_depth left open: the branch ran, found no row and changed nothing
        pre = f"{tag}_i{iters}_"
        if iters == 1:
            gp = T(g[pre + "g_poses"])
            assert torch.allclose(win.g_pose.cpu(), gp, rtol=2e-3, atol=2e-4 * float(gp.abs().max())), float((win.g_pose.cpu() - gp).abs().max() / gp.abs().max())
            # (the table gradients were consumed by Adam; beta's sits in the decoder segment Adam clears -> compare through the state)
        np.testing.assert_allclose(win.poses.cpu().numpy(), g[pre + "poses"], rtol=0, atol=2e-5)
        out = win.c2ws().cpu()
        np.testing.assert_allclose(out[-1].numpy(), g[pre + "cur_c2w"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(out[:-1].numpy(), g[pre + "kf_c2w"][frames], rtol=0, atol=3e-5)
        if iters == 2:
            _tables_after_adam(es.params.detach().cpu().numpy(), g[pre + "grid_s"], cancel=cancel_s)
            _tables_after_adam(ec.params.detach().cpu().numpy(), g[pre + "grid_c"], cancel=cancel_c)
            for k, v in dec.state_dict().items():
                np.testing.assert_allclose(v.cpu().numpy(), g[pre + "dec__" + k.replace(".", "__")], rtol=1e-3, atol=2e-5)


def test_mapwindow_first_iteration_gradients_against_reference(golden):
    """the table and beta gradients of g14's first iteration (forward + backward of the window, no optimiser step)"""
    import unislam_amd as us
    g = golden("g14_mapping_joint")
    dec, es, ec = _g14_scene(us, g)
    c2ws, depths, colors, dirs, n_per, extra, frames = g14_window(g, "w6")
    b, P = depths.shape
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=b * n_per)
    win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, cam_lr=float(g["cam_lr"]))
    idx = torch.randint(P, (n_per * b,)).reshape(b, -1)
    ro, rd, gd, _ = O.window_rays(c2ws[0], win.poses.cpu(), depths, colors, dirs, idx)
    inside = O.bbox_far(ro, rd, BOUND) >= gd
    t_rand = torch.zeros(ro.shape[0], 40); t_rand[inside] = torch.rand(int(inside.sum()), 40)
    win.draw(idx.to(DEV))
    step.lr = {k: 0.0 for k in step.lr}; win.cam_lr = 0.0
    step.forward(*win.rays(), t_rand.to(DEV), False)
    step.backward(ray_grads=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(es.params.grad.cpu().numpy(), g["w6_i1_g_grid_s"], rtol=1e-3, atol=1e-5 * float(np.abs(g["w6_i1_g_grid_s"]).max()))
    np.testing.assert_allclose(ec.params.grad.cpu().numpy(), g["w6_i1_g_grid_c"], rtol=1e-3, atol=1e-5 * float(np.abs(g["w6_i1_g_grid_c"]).max()))
    np.testing.assert_allclose(float(dec.beta.grad.reshape(-1)[0]), float(np.asarray(g["w6_i1_g_beta"]).reshape(-1)[0]), rtol=1e-3)


@pytest.mark.parametrize("holes", [False, True])
def test_mapwindow_graph_replay_equals_eager(holes):
    """capture() / replay(): five replayed joint_opt iterations == five eager ones on the same draws and jitter (loss 1e-6, parameters
    and poses allclose); capturing leaves model, optimiser and poses alone; the extra-ray block rides along.  holes: every seventh pool
    pixel has no depth -- the importance-sampling branch of src/utils/Renderer.py:104-130 is part of the captured graph (its row count
    is read on the device), with its own static draws"""
    import unislam_amd as us
    b, P, n_per, extra = 8, 500, 64, (10, 25)
    c2ws, depths, colors, dirs = _window(b, P, 7)
    if holes:
        depths[:, ::7] = 0.0
    g = torch.Generator().manual_seed(5)
    R = b * n_per + min(extra[0], b) * extra[1]
    draws = [(torch.randint(P, (b, n_per), generator=g).to(DEV), torch.randint(P, (min(extra[0], b), extra[1]), generator=g).to(DEV),
              torch.rand(R, 40, generator=g).to(DEV)) for _ in range(6)]
    zds = [(torch.rand(R, 32, generator=g).to(DEV), torch.rand(R, 8, generator=g).to(DEV)) if holes else None for _ in range(6)]
    outs = []
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
        win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, cam_lr=1e-3, extra=extra, has_zero_depth=None if holes else False)
        losses = [float(win.iterate(draws[0][0], draws[0][1], t_rand=draws[0][2], zero_depth_draws=zds[0]))]   # one eager step first: moments are non-zero
        if holes:
            assert 20 < int(step.zd_count) < R // 4
        if mode == "graph":
            before = (step.flat.clone(), step.m.clone(), win.poses.clone(), win.pm.clone(), float(step.step_dev[0]))
            win.capture(t_rand=True, device_draw=False)
            assert torch.equal(step.flat, before[0]) and torch.equal(step.m, before[1]) and torch.equal(win.poses, before[2])
            assert torch.equal(win.pm, before[3]) and float(step.step_dev[0]) == before[4] == 1.0
        for (ia, ib, tr), zd in zip(draws[1:], zds[1:]):
            if mode == "graph":
                win.t_rand.copy_(tr)
                if holes:
                    win.zd_draws[0].copy_(zd[0]); win.zd_draws[1].copy_(zd[1])
                losses.append(float(win.replay(ia, ib)))
            else:
                losses.append(float(win.iterate(ia, ib, t_rand=tr, zero_depth_draws=zd)))
        assert float(step.step_dev[0]) == 6.0
        outs.append((losses, step.flat.clone(), win.poses.clone()))
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-6)
    assert torch.allclose(outs[1][1], outs[0][1], rtol=1e-6, atol=1e-8)
    assert torch.allclose(outs[1][2], outs[0][2], rtol=0, atol=1e-7)
    assert float((outs[0][2] - O.matrix_to_cam_pose(c2ws[1:]).to(DEV)).abs().max()) > 1e-3       # the poses did move


def test_fused_tracking_chain_equals_general_chain():
    """TrackStep.iterate_fused: the nine-launch chain (us_track_sample, us_hashgrid_fwd_joint_dydx, us_track_loss_fwd / _bwd,
    us_hashgrid_dydx_rays, us_pose_window_step) against the general chain on the same pixels and jitter, over four iterations"""
    import unislam_amd as us
    torch.manual_seed(3)
    dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
    for p in dec.parameters():
        p.requires_grad_(False)
    es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
    with torch.no_grad():
        es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
    H, Wd, fx, fy, cx, cy, eh, ew, n = 60, 80, 40.0, 40.0, 39.5, 29.5, 4, 5, 300
    g = torch.Generator().manual_seed(4)
    gt_depth = (torch.rand(H, Wd, generator=g) * 1.5 + 0.5).to(DEV); gt_depth[20, 20:30] = 0.0; gt_depth[30:34, 40:50] = 60.0
    gt_color = torch.rand(H, Wd, 3, generator=g).to(DEV)
    w = dict(fs=10, center=200, tail=50, color=5, depth=1)
    pose0 = torch.tensor([0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0], device=DEV)
    draws = [(torch.randint((H - 2 * eh) * (Wd - 2 * ew), (n,), generator=g).to(DEV), torch.rand(n, 40, generator=g).to(DEV)) for _ in range(4)]
    steps = []
    for fast in (True, False):
        ts = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
        ts.fast_path = fast
        ts.begin_frame(pose0, gt_color, gt_depth, 2e-3, 1e-3, H, Wd, fx, fy, cx, cy, eh, ew)
        out = []
        for idx, tr in draws:
            loss, unc, valid = ts.iterate_fused(n, t_rand=tr, indices=idx)
            out.append((float(loss), unc.clone(), valid.clone(), ts.g_pose.clone(), ts.pose.clone(), float(ts.median), ts.stats.clone()))
        steps.append(out)
    for a, b in zip(*steps):
        assert torch.equal(a[2], b[2]) and 0 < int(a[2].sum()) < n                       # some rays dropped by the pre-filter
        assert a[5] == b[5]                                                              # the median is an element of the batch: exact
        np.testing.assert_allclose(a[0], b[0], rtol=1e-6)
        assert torch.allclose(a[6], b[6], rtol=1e-6) and torch.allclose(a[1], b[1], rtol=1e-6, atol=1e-9)
        assert torch.allclose(a[3], b[3], rtol=1e-4, atol=1e-5 * float(b[3].abs().max()))      # (dL/d(point) summed in another order)
        assert torch.allclose(a[4], b[4], rtol=0, atol=1e-6)


def test_track_sample_draws_its_pixels_inside_the_crop():
    """us_track_sample with pix = NULL: the in-kernel draw stays inside the crop, covers it evenly, changes with the device-side step
    count, and the gathered depth / colour / direction belong to the drawn pixel"""
    import unislam_amd as us
    from unislam_amd import _lib as L
    lib, P_ = L.lib(), L.ptr
    H, Wd, fx, fy, cx, cy, eh, ew, n = 60, 80, 40.0, 41.0, 39.5, 29.5, 4, 5, 8192
    g = torch.Generator().manual_seed(1)
    depth = (torch.rand(H, Wd, generator=g) + 0.5).to(DEV); color = torch.rand(H, Wd, 3, generator=g).to(DEV)
    pose = torch.tensor([0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0], device=DEV)
    t_uni, t_surf = torch.linspace(0, 1, 32).to(DEV), torch.linspace(0, 1, 8).to(DEV)
    f = lambda *s: torch.empty(s, device=DEV)
    outs = []
    for step in (0.0, 1.0):
        ctr = torch.tensor([step], device=DEV)
        ro, rd, dirs, gd, gc, z, pts = f(n, 3), f(n, 3), f(n, 3), f(n), f(n, 3), f(n, 40), f(n, 40, 3)
        valid = torch.empty(n, dtype=torch.uint8, device=DEV)
        L.check(lib.us_track_sample(P_(pose), None, n, L.host_floats([fx, fy, cx, cy]), ew, eh, Wd - 2 * ew, H - 2 * eh, P_(depth), P_(color), Wd,
                                    us.common.bound_host(BOUND), P_(t_uni), 32, P_(t_surf), 8, L.ctypes.c_float(1.2), L.ctypes.c_float(0.09),
                                    L.ctypes.c_float(0.18), None, 1234, P_(ctr), 1, P_(ro), P_(rd), P_(dirs), P_(gd), P_(gc), P_(valid), P_(z),
                                    P_(pts), L.stream()), "us_track_sample")
        u = torch.round(dirs[:, 0] * fx + cx).long(); v = torch.round(-dirs[:, 1] * fy + cy).long()
        assert int(u.min()) >= ew and int(u.max()) < Wd - ew and int(v.min()) >= eh and int(v.max()) < H - eh
        assert torch.equal(gd, depth[v, u]) and torch.equal(gc, color[v, u])
        assert abs(float((u < Wd // 2).float().mean()) - 0.5) < 0.03 and abs(float((v < H // 2).float().mean()) - 0.5) < 0.03
        assert len(torch.unique(v * Wd + u)) > 0.6 * (H - 2 * eh) * (Wd - 2 * ew)          # 8192 draws over 3640 pixels: most are hit
        c2w = us.common.cam_pose_to_matrix(pose[None])
        o_ref, d_ref = us.common.get_rays_from_uv(u.float()[None], v.float()[None], c2w, H, Wd, fx, fy, cx, cy, DEV)
        assert torch.allclose(rd, d_ref.reshape(-1, 3), rtol=1e-5, atol=1e-6) and torch.allclose(ro, o_ref.reshape(-1, 3))
        assert bool((z[:, 1:] >= z[:, :-1]).all())
        outs.append(v * Wd + u)
    assert not torch.equal(outs[0], outs[1])


def test_window_sample_equals_rays_then_sample_points():
    """us_window_sample (pose -> rays -> pre-filter, z, points in one launch) == us_window_rays (both blocks) + us_sample_points, bit for
    bit, on given indices and jitter; with NULL indices it draws its pixels inside the pools and differently per device-side step count"""
    import unislam_amd as us
    from unislam_amd import _lib as L
    for b, n_per, extra in ((6, 20, None), (12, 10, (10, 37)), (3, 50, (10, 8)), (1, 64, None)):
        c2ws, depths, colors, dirs = _window(b, 300, 80 + b)
        torch.manual_seed(0)
        dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(12)).to(DEV), us.HashGridEncoding(3, _ecfg(12)).to(DEV)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=64)
        win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=b > 1, extra=extra, has_zero_depth=False)
        g = torch.Generator().manual_seed(b)
        nf = win.extra[0] if win.extra else 0
        idx = torch.randint(300, (b, n_per), generator=g).to(DEV)
        idx2 = torch.randint(300, (nf, extra[1]), generator=g).to(DEV) if extra else None
        t_rand = torch.rand(win.R, 40, generator=g).to(DEV)
        # reference: the two-step path
        win.draw(idx, idx2)
        ro, rd, gd, gc = [t.clone() for t in win.rays()]
        dirs_ref = win.dirs.clone()
        step.forward(ro, rd, gd, gc, t_rand, False, backward_follows=False)
        ref = (step.valid[:win.R].clone(), step.z[:win.R].clone(), step.pts[:win.R].clone())
        for t in (win.ro, win.rd, win.gd, win.gc, win.dirs, step.z, step.pts):
            t.zero_()
        step.valid.zero_()
        step.lr = {k: 0.0 for k in step.lr}; win.cam_lr = 0.0
        win.iterate(idx, idx2, t_rand=t_rand)
        for a, r in zip((win.ro, win.rd, win.gd, win.gc, win.dirs, step.valid[:win.R], step.z[:win.R], step.pts[:win.R]), (ro, rd, gd, gc, dirs_ref) + ref):
            assert torch.equal(a, r), (b, extra)
        # in-kernel draw
        win.iterate()
        g1 = win.gd.clone()
        pool = win.pool_d
        rows = torch.arange(b, device=DEV).repeat_interleave(n_per)
        if extra:
            rows = torch.cat([rows, (b - nf + torch.arange(nf, device=DEV)).repeat_interleave(extra[1])])
        assert bool(((pool[rows] - g1[:, None]).abs().min(dim=1)[0] == 0).all())              # every drawn depth is in its frame's pool
        win.iterate()
        assert not torch.equal(win.gd, g1)                                                      # the step count advanced: another draw


def test_mapwindow_graph_with_device_side_draw():
    """capture() (default): the replayed graph draws its own pixels -- replays see different batches, the optimiser advances, no host call
    between them but the graph launch"""
    import unislam_amd as us
    b, P, n_per = 8, 500, 64
    c2ws, depths, colors, dirs = _window(b, P, 9)
    torch.manual_seed(0)
    dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=b * n_per)
    win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, has_zero_depth=False)
    win.capture()
    with pytest.raises(us.UniSlamHipError):
        win.replay(torch.zeros(b, n_per, dtype=torch.int64, device=DEV))
    seen, losses = [], []
    for _ in range(4):
        losses.append(float(win.replay()))
        seen.append(win.gd.clone())
    assert all(np.isfinite(losses)) and float(step.step_dev[0]) == 4.0
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


@pytest.mark.parametrize("S,n_rays", [(64, 70), (40, 133)])
def test_decoder_backward_contracts_dydx_in_registers(S, n_rays):
    """us_mlp_bwd_pair_dydx + us_ray_points_bwd2 == us_mlp_bwd_pair + us_hashgrid_dydx_rays (another summation order: 1e-5), with and
    without the dL/d(features) output and the parameter gradients; dL/d(features) and the parameter gradients themselves unchanged"""
    import ctypes
    import unislam_amd as us
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    N = n_rays * S
    g = torch.Generator().manual_seed(S)
    ds_ = us.make_mlp_desc(32, 32, 2, 1, "tanh", True, "bf16"); dc_ = us.make_mlp_desc(32, 32, 2, 3, "sigmoid", True, "bf16")
    A, B = ctypes.byref(ds_), ctypes.byref(dc_)
    ps = (torch.randn(us.network.mlp_n_params(ds_), generator=g) * 0.3).to(DEV); pc = (torch.randn(us.network.mlp_n_params(dc_), generator=g) * 0.3).to(DEV)
    fa, fb = torch.randn(16, N, 2, generator=g).to(DEV), torch.randn(16, N, 2, generator=g).to(DEV)
    dda, ddb = torch.randn(16, 3, N, 2, generator=g).half().to(DEV), torch.randn(16, 3, N, 2, generator=g).half().to(DEV)
    d_raw = torch.randn(N, 4, generator=g).to(DEV); z = (torch.rand(n_rays, S, generator=g) * 3).to(DEV)
    raw = torch.empty(N, 4, device=DEV)
    off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
    L.check(lib.us_mlp_fwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), N, off(raw, 3), 4, P(raw), 4, 1, st), "fwd")
    f = lambda *s: torch.zeros(s, device=DEV)
    wsb = int(lib.us_mlp_bwd_workspace_bytes(A))
    bh = us.common.bound_host(BOUND)

    def run(fused, want_din, want_w):
        da, db = (f(16, N, 2), f(16, N, 2)) if (want_din or not fused) else (None, None)
        gs, gc = (torch.zeros_like(ps), torch.zeros_like(pc)) if want_w else (None, None)
        wa, wb = (torch.empty(wsb, dtype=torch.uint8, device=DEV), torch.empty(wsb, dtype=torch.uint8, device=DEV)) if want_w else (None, None)
        go, gd = f(n_rays, 3), f(n_rays, 3)
        common = (A, B, P(ps), P(pc), P(fa), P(fb), off(raw, 3), 4, P(raw), 4, off(d_raw, 3), 4, P(d_raw), 4, N, P(da), P(db), P(gs), P(gc), 1,
                  P(wa), P(wb), wsb if want_w else 0)
        if fused:
            pa_, pb_ = f(N, 3), f(N, 3)
            L.check(lib.us_mlp_bwd_pair_dydx(*common, P(dda), P(ddb), P(pa_), P(pb_), st), "us_mlp_bwd_pair_dydx")
            L.check(lib.us_ray_points_bwd2(P(pa_), P(pb_), P(z), bh, n_rays, S, P(go), P(gd), st), "us_ray_points_bwd2")
        else:
            L.check(lib.us_mlp_bwd_pair(*common, st), "us_mlp_bwd_pair")
            L.check(lib.us_hashgrid_dydx_rays(16, P(da), P(db), P(dda), P(ddb), n_rays, S, P(z), bh, P(go), P(gd), None, st), "us_hashgrid_dydx_rays")
        return go, gd, da, db, gs, gc

    ref = run(False, True, True)
    for want_din, want_w in ((True, True), (False, False), (True, False)):
        out = run(True, want_din, want_w)
        for a, b in zip(out[:2], ref[:2]):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max())), float((a - b).abs().max() / b.abs().max())
        if want_din:
            assert torch.equal(out[2], ref[2]) and torch.equal(out[3], ref[3])
        if want_w:
            assert torch.equal(out[4], ref[4]) and torch.equal(out[5], ref[5])


def test_mapwindow_with_pixels_without_depth():
    """a window whose pools hold pixels without a depth measurement (src/utils/Renderer.py:104-130: the importance-sampling branch):
    MapWindow gives the loss statistics and the pose gradient input of MapStep on the same rays, jitter and draws (the branch's row
    count stays on the device: us_zero_depth_resample)"""
    import unislam_amd as us
    b, P, n_per = 5, 400, 60
    c2ws, depths, colors, dirs = _window(b, P, 21)
    depths[:, ::7] = 0.0
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(P, (b, n_per), generator=g).to(DEV)
    R = b * n_per
    t_rand = torch.rand(R, 40, generator=g).to(DEV)
    n0 = int((depths.to(DEV).gather(1, idx) <= 0).sum())
    assert n0 > 10
    zd = (torch.rand(n0, 32, generator=g).to(DEV), torch.rand(n0, 8, generator=g).to(DEV))
    outs = []
    for mode in ("window", "step"):
        torch.manual_seed(0)
        dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
        win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True)
        assert win.has_zero
        if mode == "window":
            step.lr = {k: 0.0 for k in step.lr}; win.cam_lr = 0.0
            loss = win.iterate(idx, t_rand=t_rand, zero_depth_draws=zd)
            g_o, g_d = step.g_o[:R].clone(), step.g_d[:R].clone()
        else:
            win.draw(idx)
            loss = step.forward_backward(*win.rays(), t_rand=t_rand, has_zero_depth=True, ray_grads=True, zero_depth_draws=zd)
            g_o, g_d = [t.clone() for t in step.ray_gradients()]
        outs.append((float(loss), step.stats.clone(), step.z[:R].clone(), g_o, g_d))
    a, c = outs
    assert np.isfinite(a[0]) and a[0] == c[0] and torch.equal(a[1], c[1]) and torch.equal(a[2], c[2])
    for k in (3, 4):                                               # (the window contracts stored half-precision dy/dx, the step re-gathers in fp32)
        assert torch.allclose(a[k], c[k], rtol=2e-3, atol=2e-4 * float(c[k].abs().max()))


def test_a_captured_iteration_does_not_keep_its_owner_in_a_cycle():
    """hipGraphDestroy is not permitted while a stream captures: a graph that is only reachable from a garbage cycle would be destroyed by
    the cyclic collector at an arbitrary later time -- e.g. inside the NEXT capture (a 40-frame SLAM run with one captured MapWindow
    per mapped frame aborted there).  The owner of a CapturedIteration must therefore die by reference counting alone."""
    import gc, weakref
    from unislam_amd.graph import CapturedIteration

    class Owner:
        def __init__(self):
            self.x = torch.zeros(8, device=DEV)
            self.it = CapturedIteration(lambda: self.work(), warmup=1)

        def work(self):
            self.x.add_(1.0)
            return self.x

    was = gc.isenabled()
    gc.disable()
    try:
        o = Owner()
        before = float(o.x[0])
        o.it.replay(); torch.cuda.synchronize()
        assert float(o.x[0]) == before + 1.0 and o.it.fn is None
        r = weakref.ref(o)
        del o
        assert r() is None                                                 # no collector involved
    finally:
        if was:
            gc.enable()


def test_graph_registry_destroys_nothing_while_a_graph_is_alive():
    """graph.py's release rule on real graphs (each MapWindow graph forks side streams): while one captured graph is alive nothing is
    destroyed, whatever the order the owners die in -- destroying a graph NEWER than a living one is the fault of
    profiles/r05_hipgraph_destroy_segv.txt, destroying OLDER ones and then capturing again is the one this test used to end in
    (profiles/r06_graph_release_rules.txt); once every owner is gone the registry empties before the next capture, and that capture
    replays.  tools/graph_fifo_check.py / graph_fifo_check2.py are the sequences that looked safe under the first rule."""
    import gc
    import unislam_amd as us
    from unislam_amd import graph

    def make(seed):
        torch.manual_seed(seed)
        dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        c2ws, depths, colors, dirs = _window(4, 300, seed)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=4 * 48)
        win = us.MapWindow(step, c2ws, depths, colors, dirs, 48, joint_opt=True, cam_lr=1e-3, has_zero_depth=False)
        win.capture()
        return win

    assert os.environ.get("US_GRAPH_RELEASE", "idle") == "idle"
    gc.collect(); graph.collect()
    n0 = len(graph._KEEP)                                       # graphs of earlier tests that wait for an owner that is still alive
    a, b, c = make(1), make(2), make(3)
    assert len(graph._KEEP) == n0 + 3
    del b; gc.collect()
    assert graph.collect() == 0 and len(graph._KEEP) == n0 + 3
    for _ in range(5):
        a.replay(); c.replay()
    del a; gc.collect()
    assert graph.collect() == 0 and len(graph._KEEP) == n0 + 3  # c lives: a's and b's graphs wait
    for _ in range(10):
        la = c.replay()
    torch.cuda.synchronize()
    assert np.isfinite(float(la))
    d = make(4)                                                 # (a capture runs collect() itself)
    assert len(graph._KEEP) == n0 + 4
    for _ in range(5):
        d.replay(); c.replay()
    torch.cuda.synchronize()
    del c, d, la; gc.collect()
    if n0 == 0:
        assert graph.collect() == 4 and graph._KEEP == []       # no owner left: all four go, then a fresh capture replays
        e = make(5)
        for _ in range(5):
            le = e.replay()
        torch.cuda.synchronize()
        assert np.isfinite(float(le)) and len(graph._KEEP) == 1


def test_track_step_keeps_the_minimum_loss_pose_and_mean_uncertainty():
    """us_pose_track_step: the pose step with the tracker's minimum-loss bookkeeping in its launch (src/Tracker.py:346-348) against
    us_pose_window_step + the same bookkeeping on torch ops, both chains, bit for bit; us_masked_mean against torch (:353)"""
    import unislam_amd as us
    from unislam_amd import _lib as L
    torch.manual_seed(3)
    dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
    for p in dec.parameters():
        p.requires_grad_(False)
    es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
    with torch.no_grad():
        es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
    H, Wd, fx, fy, cx, cy, eh, ew, n = 60, 80, 40.0, 40.0, 39.5, 29.5, 4, 5, 300
    g = torch.Generator().manual_seed(4)
    gt_depth = (torch.rand(H, Wd, generator=g) * 1.5 + 0.5).to(DEV); gt_depth[20, 20:30] = 0.0
    gt_color = torch.rand(H, Wd, 3, generator=g).to(DEV)
    w = dict(fs=10, center=200, tail=50, color=5, depth=1)
    pose0 = torch.tensor([0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0], device=DEV)
    draws = [(torch.randint((H - 2 * eh) * (Wd - 2 * ew), (n,), generator=g).to(DEV), torch.rand(n, 40, generator=g).to(DEV)) for _ in range(8)]
    for fast in (True, False):
        res = []
        for keep in (True, False):                                                       # (both runs keep the candidate; the second ignores it)
            ts = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
            ts.fast_path = fast
            ts.begin_frame(pose0, gt_color, gt_depth, 2e-2, 1e-2, H, Wd, fx, fy, cx, cy, eh, ew)      # large steps: the loss goes up and down
            min_loss, cand, losses = torch.full((1,), float("inf"), device=DEV), ts.pose.clone(), []
            for idx, tr in draws:
                before = ts.pose.clone()
                loss, unc, valid = ts.iterate_fused(n, t_rand=tr, indices=idx)
                better = loss < min_loss
                min_loss, cand = torch.where(better, loss, min_loss), torch.where(better, before, cand)
                losses.append(float(loss))
            res.append((losses, ts.pose.clone(), min_loss, cand, ts.min_loss.clone(), ts.best_pose.clone()))
            if keep:
                m = ts.mean_uncertainty(unc, valid)
                ref = (unc * valid.float()).sum() / valid.float().sum().clamp(min=1)
                np.testing.assert_allclose(float(m), float(ref), rtol=1e-6)
        (la, pa, ma, ca, dev_min, dev_best), (lb, pb, mb, cb, _, _) = res
        assert la == lb and torch.equal(pa, pb)                                          # the step itself is unchanged
        assert min(la) < la[0] and la.index(min(la)) not in (0, len(la) - 1)             # the minimum sits inside the loop
        assert torch.equal(dev_min, ma) and torch.equal(dev_best, ca) and torch.equal(ca, cb)
        assert float(ts.draw_ctr) == len(draws) and float(ts.pstep) == len(draws)         # the draw's counter advances with the step ...
        ts.begin_frame(pose0, gt_color, gt_depth, 2e-2, 1e-2, H, Wd, fx, fy, cx, cy, eh, ew)
        assert float(ts.draw_ctr) == len(draws) and float(ts.pstep) == 0                  # ... and is not reset with the frame
    # a NaN loss never replaces the candidate; an empty selection has mean 0
    lib, P_ = L.lib(), L.ptr
    z3 = torch.zeros(4, 3, device=DEV)
    pose, m7, v7, stp = pose0.clone(), torch.zeros(7, device=DEV), torch.zeros(7, device=DEV), torch.zeros(1, device=DEV)
    mn, best = torch.tensor([0.5], device=DEV), torch.full((7,), 9.0, device=DEV)
    for val, taken in ((float("nan"), False), (0.75, False), (0.25, True)):
        lv = torch.tensor([val], device=DEV)
        L.check(lib.us_pose_track_step(P_(pose), P_(z3), P_(z3), P_(z3), 4, P_(m7), P_(v7), None, 1e-3, 1e-3, 0.5, 0.999, 1e-8, P_(stp),
                                       P_(lv), P_(mn), P_(best), None, L.stream()), "us_pose_track_step")
        assert (float(mn) == val) == taken and (float(best[4]) == 3.0) == taken
    out, vals, none = torch.ones(1, device=DEV), torch.rand(10, device=DEV), torch.zeros(10, dtype=torch.uint8, device=DEV)
    L.check(lib.us_masked_mean(P_(vals), P_(none), 10, P_(out), L.stream()), "us_masked_mean")
    assert float(out) == 0.0
    L.check(lib.us_masked_mean(P_(vals), None, 10, P_(out), L.stream()), "us_masked_mean")
    np.testing.assert_allclose(float(out), float(vals.double().mean()), rtol=1e-6)


def test_pose_conversion_kernels_equal_the_torch_chains():
    """us_matrix_to_cam_pose / us_cam_pose_to_matrix (one launch each) against the torch restatement of pytorch3d's matrix_to_quaternion /
    quaternion_to_matrix (src/common.py:182-208) -- every branch of the candidate selection (rotations by up to 180 degrees about the
    three axes and about random ones), the constant-speed extrapolation (src/Tracker.py:317-320), and the round trip"""
    import unislam_amd as us
    from unislam_amd import common as C
    g = torch.Generator().manual_seed(7)
    q = torch.randn(400, 4, generator=g)
    q[:40, 0] *= 1e-3                                                           # nearly half turns: the real part is NOT the largest
    q[40:50] = torch.eye(4)[[1, 2, 3, 0, 1, 2, 3, 1, 2, 3]] + 1e-4 * torch.randn(10, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    R = O.quaternion_to_matrix(q)
    c2w = torch.eye(4).repeat(400, 1, 1); c2w[:, :3, :3] = R; c2w[:, :3, 3] = torch.randn(400, 3, generator=g) * 3
    ref = O.matrix_to_cam_pose(c2w)                                              # CPU torch chain (oracle restatement, pinned by g14)
    got = C.matrix_to_cam_pose(c2w.to(DEV))
    assert got.shape == (400, 7) and len(set(ref[:, :4].abs().argmax(-1).tolist())) == 4       # all four branches were taken
    assert torch.allclose(got.cpu(), ref, rtol=0, atol=2e-7) and torch.equal(got[:, 4:].cpu(), c2w[:, :3, 3])
    back = C.cam_pose_to_matrix(got)
    ref_back = torch.eye(4).repeat(400, 1, 1); ref_back[:, :3, :3] = O.quaternion_to_matrix(ref[:, :4]); ref_back[:, :3, 3] = ref[:, 4:]
    assert torch.allclose(back.cpu(), ref_back, rtol=0, atol=5e-7) and torch.allclose(back.cpu(), c2w, rtol=0, atol=2e-6)
    assert torch.equal(back[:, 3].cpu(), torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(400, 1))
    # the constant-speed prediction: element-wise on the 7 numbers (Tracker.py:317-320), the newer quaternion first put on the older one's
    # hemisphere (q and -q are one rotation; r5)
    for a, b_ in ((3, 4), (100, 101), (200, 201), (45, 46)):
        pred = C.predict_cam_pose(c2w[a].to(DEV), c2w[b_].to(DEV))
        q1 = ref[b_:b_ + 1].clone()
        if float((ref[a, :4] * q1[0, :4]).sum()) < 0:
            q1[:, :4] = -q1[:, :4]
        assert pred.shape == (1, 7) and torch.allclose(pred.cpu(), 2 * q1 - ref[a:a + 1], rtol=0, atol=5e-7)
    # two poses a small turn apart on either side of a quarter turn about -x: quaternions (cos, -sin, 0, 0) with |cos| ~ |sin|; the
    # largest-candidate rule of matrix_to_quaternion switches from the r branch to the i branch between them and answers with (nearly)
    # OPPOSITE quaternions; the element-wise extrapolation without the hemisphere step would predict ~3 q instead of ~q
    import math
    def rot_x(a_):
        m = torch.eye(4); m[1, 1] = m[2, 2] = math.cos(a_); m[1, 2] = -math.sin(a_); m[2, 1] = math.sin(a_); return m
    m0, m1 = rot_x(-(math.pi / 2 - 0.01)), rot_x(-(math.pi / 2 + 0.01))
    p0, p1 = O.matrix_to_cam_pose(torch.stack([m0, m1]))
    assert float((p0[:4] * p1[:4]).sum()) < -0.99                           # the case at hand
    for dev_path in (True, False):
        pred = C.predict_cam_pose(m0.to(DEV), m1.to(DEV)) if dev_path else C.predict_cam_pose(m0.to(DEV).requires_grad_(True), m1.to(DEV))
        Rp = C.cam_pose_to_matrix(pred.detach())[0, :3, :3].cpu()
        want = rot_x(-(math.pi / 2 + 0.03))[:3, :3]                          # the same turn once more
        assert float((Rp - want).abs().max()) < 1e-3, (dev_path, Rp)
    # tensors inside autograd keep the differentiable torch chain
    p = got[:2].clone().requires_grad_(True)
    C.cam_pose_to_matrix(p).sum().backward()
    assert p.grad is not None and float(p.grad.abs().sum()) > 0


@pytest.mark.parametrize("arena", [False, True])
def test_pose_group_rides_in_the_optimiser_launch(arena):
    """joint_opt window with the bf16 decoder pair: the camera poses' group (and the decoder group) as leading workgroups of the tables'
    optimiser launch (us_adam_step_model + us_pose_step_desc) == the separate launches (us_pose_window_step / us_arena_pose_step,
    us_mlp_reduce_pair_adam, us_adam_step_segments_dev): poses, their moments, decoders and beta bit for bit, tables to the rounding of their
    f64 sums; four iterations on the same draws, with the extra rays of the newest frames"""
    import unislam_amd as us
    b, P, n_per, extra = 7, 400, 48, (10, 20)
    c2ws, depths, colors, dirs = _window(b, P, 11)
    g = torch.Generator().manual_seed(9)
    R = b * n_per + min(extra[0], b) * extra[1]
    draws = [(torch.randint(P, (b, n_per), generator=g).to(DEV), torch.randint(P, (min(extra[0], b), extra[1]), generator=g).to(DEV),
              torch.rand(R, 40, generator=g).to(DEV)) for _ in range(4)]
    outs = []
    for one_launch in (True, False):
        torch.manual_seed(0)
        dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R, deterministic=True)
        step.one_launch_adam = one_launch
        if arena:
            ar = us.KeyframeArena(b + 2, P, DEV)
            rows = []
            for f in range(b):
                r = ar.alloc() if f else 0
                ar.put(r, colors[f], depths[f], dirs[f])
                rows.append(r)
            win = us.ArenaWindow(step, ar, b * n_per, min(extra[0], b) * extra[1], joint_opt=True, cam_lr=1e-3, has_zero_depth=False)
            win.bind(rows, c2ws, n_per, extra)
            it = lambda ia, ib, tr: win.iterate(ia, ib, t_rand=tr)
        else:
            win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, cam_lr=1e-3, extra=extra, has_zero_depth=False)
            it = lambda ia, ib, tr: win.iterate(ia, ib, t_rand=tr)
        assert step._decoder_pair()
        losses = [float(it(ia, ib, tr)) for ia, ib, tr in draws]
        assert float(step.step_dev[0]) == 4.0
        outs.append((losses, step.flat.clone(), step.m.clone(), step.v.clone(), win.poses.clone(), win.pm.clone(), win.pv.clone()))
    a, c = outs
    assert a[0] == c[0]
    nd = step.o_tab_s
    for k in (1, 2, 3):
        assert torch.equal(a[k][:nd], c[k][:nd]), k
        assert torch.allclose(a[k][nd:], c[k][nd:], rtol=1e-6, atol=1e-12), k
    for k in (4, 5, 6):
        assert torch.equal(a[k], c[k]), k
    assert float(a[5].abs().max()) > 0                                          # the poses' moments did move


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_handovers_between_launches_change_no_bit(prec):
    """The two r6 hand-overs between the launches of a mapping iteration -- features written by the encoder as the split-bf16 decoders' hi / lo
    operand pairs (US_GRID_FEAT_SPLIT_BF16 / US_MLP_IN_SPLIT_BF16), output activations and their derivatives evaluated by the compositing
    launches instead of the decoder launches (US_MLP_OUT_PREACT / US_MLP_DOUT_PREACT / US_RENDER_ACT) -- against the same iteration without
    them.  First iteration (same tables on both sides): raw, rendered values, loss and the decoder gradients BIT FOR BIT, the table gradients
    to the order of the f64 sums inside a bin; dL/d(raw) is the other side's times act'(raw), bit for bit.  Later iterations start from
    tables that differ in a last bit in a handful of entries -- as between any two runs of one build -- so they are compared to 1e-6."""
    import unislam_amd as us
    b, P, n_per = 6, 500, 128
    c2ws, depths, colors, dirs = _window(b, P, 41)
    g = torch.Generator().manual_seed(6)
    draws = [(torch.randint(P, (b, n_per), generator=g).to(DEV), torch.rand(b * n_per, 40, generator=g).to(DEV)) for _ in range(3)]
    outs = []
    for on in (False, True):
        torch.manual_seed(0)
        dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": prec}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(16)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=b * n_per, deterministic=True)
        step.feat_split, step.act_handover = on, on
        win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, cam_lr=1e-3, has_zero_depth=False)
        rec = []
        for idx, tr in draws:
            loss = win.iterate(idx, None, t_rand=tr)
            rec.append((float(loss), step.raw[:win.R].clone(), step.depth[:win.R].clone(), step.rgb[:win.R].clone(), step.d_raw[:win.R].clone(),
                        step.grad.clone()))
        assert (step._ms != 0) == (on and prec == "bf16") and (step._mo != 0) == on            # the paths were taken / left as asked
        outs.append((rec, step.flat.clone(), win.poses.clone(), step.o_tab_s))
    (ra, fa, pa, o_tab), (rb, fb, pb, _) = outs

    def where(x, y):
        bad = (x != y).nonzero()
        return (len(bad), bad[:4].tolist(), x[x != y][:4].tolist(), y[x != y][:4].tolist())
    (la, raw_a, d_a, c_a, dr_a, g_a), (lb, raw_b, d_b, c_b, dr_b, g_b) = ra[0], rb[0]
    assert la == lb, (la, lb)
    assert torch.equal(raw_a, raw_b), where(raw_a, raw_b)
    assert torch.equal(d_a, d_b) and torch.equal(c_a, c_b), (where(d_a, d_b), where(c_a, c_b))
    # dL/d(raw): with the hand-over it is the gradient w.r.t. the PRE-activation outputs -- the other's times act'(raw)
    want = dr_a.clone()
    want[..., :3] = dr_a[..., :3] * (raw_a[..., :3] * (1.0 - raw_a[..., :3]))
    want[..., 3] = dr_a[..., 3] * (1.0 - raw_a[..., 3] * raw_a[..., 3])
    assert torch.equal(dr_b, want), where(dr_b, want)
    assert torch.equal(g_a[:o_tab], g_b[:o_tab]), where(g_a[:o_tab], g_b[:o_tab])            # decoders + beta
    assert float(g_a[o_tab:].abs().max()) > 0 and torch.allclose(g_a[o_tab:], g_b[o_tab:], rtol=1e-6, atol=1e-12)
    for it in (1, 2):
        (la, raw_a, d_a, c_a, _, g_a), (lb, raw_b, d_b, c_b, _, g_b) = ra[it], rb[it]
        assert abs(la - lb) <= 1e-6 * abs(la), (it, la, lb)
        for x, y in ((raw_a, raw_b), (d_a, d_b), (c_a, c_b)):
            assert torch.allclose(x, y, rtol=1e-6, atol=1e-7), (it, float((x - y).abs().max()))
        assert float((raw_a != raw_b).float().mean()) < 1e-3                                    # (a last bit here and there, not a different path)
    d = (fa - fb).abs()
    assert float((d > 0).float().mean()) < 1e-3 and float(d.max()) < 1e-5, (float((d > 0).float().mean()), float(d.max()))
    assert torch.allclose(pa, pb, rtol=0, atol=1e-7)
