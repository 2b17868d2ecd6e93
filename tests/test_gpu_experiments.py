"""
GPU: parity tests of the EXPERIMENTS build (include/unislam_hip_experiments.h; tools/build_experiments.sh) -- variants that were measured
slower than the shipped kernels and are kept buildable, held to the same bars: packed 8-byte records of the binned
table gradient, the one-launch encode + decode kernel for render-only calls.  With the shipped library every test here is skipped.
"""
import ctypes

import numpy as np
import pytest
import torch

import unislam_oracle as O
from test_gpu_joint import _pair, _ray_points
from test_gpu_parity import PLS816, enc_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


@pytest.fixture(scope="module")
def us():
    import unislam_amd
    from unislam_amd import _lib as L
    assert torch.cuda.is_available()
    if not L.has_experiments():
        pytest.skip("the loaded libunislam_hip.so is the shipped build (tools/build_experiments.sh builds the variants)")
    return unislam_amd


@pytest.mark.parametrize("l2a,l2b,res,n", [(16, 19, 816, 70001), (16, 16, 456, 4096), (19, 19, 816, 1), (14, 15, 816, 63)])
@pytest.mark.parametrize("width,n_hidden,bias,prec", [(32, 2, True, "bf16"), (32, 2, False, "bf16_plain"), (16, 1, True, "bf16"), (64, 2, True, "bf16"),
                                                      (64, 1, False, "bf16_plain")])
def test_encode_decode_in_one_launch_equals_the_four_launches(us, l2a, l2b, res, n, width, n_hidden, bias, prec):
    """us_encode_decode_fwd (csrc/encode_decode.inc: both grids and both decoders in one kernel, features kept in LDS) against
    us_hashgrid_fwd + us_mlp_fwd per grid / decoder on the same points and parameters: bit-identical outputs, written into one
    raw[N][4] the way the render path does (sdf -> column 3, rgb -> columns 0..2)."""
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    x, g = _ray_points(n, 300 + n)
    ea, eb = _pair(us, l2a, l2b, res, g)
    da, db = ctypes.byref(ea.desc), ctypes.byref(eb.desc)
    ma = us.make_mlp_desc(32, width, n_hidden, 1, "none", bias, prec)
    mb = us.make_mlp_desc(32, width, n_hidden, 3, "sigmoid", bias, prec)
    pa = ((torch.rand(us.network.mlp_n_params(ma), device=DEV, generator=g) * 2 - 1) * 0.4)
    pb = ((torch.rand(us.network.mlp_n_params(mb), device=DEV, generator=g) * 2 - 1) * 0.4)
    assert lib.us_encode_decode_supported(da, db, ctypes.byref(ma), ctypes.byref(mb)) == 1
    off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
    for clamp in (1, 0):
        xin = x if clamp else x.clamp(0, 1)
        fa, fb = torch.empty((16, n, 2), device=DEV), torch.empty((16, n, 2), device=DEV)
        ref, out = torch.full((n, 4), -7.0, device=DEV), torch.full((n, 4), -7.0, device=DEV)
        L.check(lib.us_hashgrid_fwd(da, P(ea.params.detach()), P(xin), n, P(fa), None, clamp | 2, st), "fwd a")
        L.check(lib.us_hashgrid_fwd(db, P(eb.params.detach()), P(xin), n, P(fb), None, clamp | 2, st), "fwd b")
        L.check(lib.us_mlp_fwd(ctypes.byref(ma), P(pa), P(fa), n, off(ref, 3), 4, 1, st), "mlp a")
        L.check(lib.us_mlp_fwd(ctypes.byref(mb), P(pb), P(fb), n, P(ref), 4, 1, st), "mlp b")
        L.check(lib.us_encode_decode_fwd(da, db, P(ea.params.detach()), P(eb.params.detach()), ctypes.byref(ma), ctypes.byref(mb), P(pa), P(pb),
                                         P(xin), n, off(out, 3), 4, P(out), 4, clamp, st), "encode_decode")
        assert torch.equal(out, ref)
    # fp32 decoders, decoders of different shapes: not this path's
    assert lib.us_encode_decode_supported(da, db, ctypes.byref(us.make_mlp_desc(32, width, n_hidden, 1, "none", bias, "fp32")), ctypes.byref(mb)) == 0
    assert lib.us_encode_decode_supported(da, db, ctypes.byref(us.make_mlp_desc(32, {16: 32, 32: 16, 64: 32}[width], n_hidden, 1, "none", bias, prec)),
                                          ctypes.byref(mb)) == 0


@pytest.mark.parametrize("log2T", [14, 19])
def test_binned_backward_packed_records(us, log2T):
    """US_GRID_BWD_PACKED: 8-byte records (values rounded to 26 / 27 significant bits) against the oracle and the unpacked pass;
    a NaN and an inf in the incoming gradient reach exactly the entries they reach unpacked; F != 2 is refused."""
    import ctypes
    from unislam_amd import _lib as L
    rng = np.random.default_rng(21)
    n = 50000
    x = rng.random((n, 3), dtype=np.float32)
    x[:20000] = (0.3 + 0.05 * rng.random((20000, 3))).astype(np.float32)             # long runs and hot bins
    dy = (rng.standard_normal((n, 32)) * np.exp(rng.uniform(-12, 4, (n, 1)))).astype(np.float32)    # 7 decades of magnitudes
    d = O.make_grid_desc(16, 2, log2T, 16, PLS816)
    enc = us.HashGridEncoding(3, enc_cfg(log2T)).to(DEV)
    gp = O.hashgrid_bwd_params(d, x, dy)
    xd, dyd = T(x).to(DEV), T(dy).to(DEV)
    lib, P, st = L.lib(), L.ptr, L.stream()
    nbytes = int(lib.us_hashgrid_bwd_workspace_bytes(ctypes.byref(enc.desc), n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    run = lambda dy_t, flags: (lambda g: (L.check(lib.us_hashgrid_bwd_binned(ctypes.byref(enc.desc), P(xd), P(dy_t), n, P(g),
                                                                            L.US_GRID_BWD_OVERWRITE | flags, P(ws), nbytes, st), "binned"), g)[1])(
        torch.full((d.n_params,), 7.0, device=DEV))
    g_exact, g_packed = run(dyd, 0), run(dyd, L.US_GRID_BWD_PACKED)
    scale = np.abs(gp).max()
    np.testing.assert_allclose(g_packed.cpu().numpy(), gp, rtol=1e-4, atol=2e-6 * scale)
    # against the unpacked pass: every record is off by <= 2^-18 relative (17 mantissa bits kept), so a sum is off by <= 2^-18 * sum |records|
    mag = torch.full((d.n_params,), 0.0, device=DEV)
    L.check(lib.us_hashgrid_bwd_binned(ctypes.byref(enc.desc), P(xd), P(dyd.abs().contiguous()), n, P(mag), L.US_GRID_BWD_OVERWRITE, P(ws), nbytes, st), "binned")
    err = (g_packed - g_exact).abs()
    assert bool((err <= (2.0 ** -18 + 2.0 ** -22) * mag + 1e-30).all())      # + the fp32 casts of the two sums, float((err / (mag + 1e-30)).max())
    assert float(err.max()) > 0                                                     # and it is a different pass
    # non-finite gradients propagate to the same entries
    dy_bad = dyd.clone(); dy_bad[123, 5] = float("nan"); dy_bad[456, 20] = float("inf"); dy_bad[789, 31] = -float("inf")
    b_exact, b_packed = run(dy_bad, 0), run(dy_bad, L.US_GRID_BWD_PACKED)
    assert torch.equal(torch.isnan(b_exact), torch.isnan(b_packed)) and torch.equal(torch.isinf(b_exact), torch.isinf(b_packed))
    assert int(torch.isnan(b_exact).sum()) > 0 and int(torch.isinf(b_exact).sum()) > 0
    # counted forward + packed
    feat = torch.empty(n, 32, device=DEV)
    L.check(lib.us_hashgrid_fwd_counted(ctypes.byref(enc.desc), P(enc.params.detach()), P(xd), n, P(feat), 0, P(ws), nbytes, st), "fwd counted")
    g_c = torch.empty(d.n_params, device=DEV)
    L.check(lib.us_hashgrid_bwd_binned(ctypes.byref(enc.desc), P(xd), P(dyd), n, P(g_c), L.US_GRID_BWD_OVERWRITE | L.US_GRID_BWD_COUNTED | L.US_GRID_BWD_PACKED,
                                       P(ws), nbytes, st), "counted packed")
    np.testing.assert_allclose(g_c.cpu().numpy(), gp, rtol=1e-4, atol=2e-6 * scale)
    # other feature widths are refused loudly
    cfg1 = dict(enc_cfg(14)); cfg1["n_features_per_level"] = 4
    enc4 = us.HashGridEncoding(3, cfg1).to(DEV)
    dy4 = torch.zeros(64, 64, device=DEV); x4 = torch.rand(64, 3, device=DEV)
    nb4 = int(lib.us_hashgrid_bwd_workspace_bytes(ctypes.byref(enc4.desc), 64))
    ws4 = torch.empty(nb4, dtype=torch.uint8, device=DEV); g4 = torch.zeros(enc4.params.numel(), device=DEV)
    rc = lib.us_hashgrid_bwd_binned(ctypes.byref(enc4.desc), P(x4), P(dy4), 64, P(g4), L.US_GRID_BWD_PACKED, P(ws4), nb4, st)
    assert rc == L.US_ERR_CONFIG


# ---- r6: moved here from the shipped suite together with their entry points: built, tested, measured slower
from test_gpu_window import _window, _cfg as _wcfg, _ecfg, BOUND, W, LR  # noqa: E402


def test_table_gradient_cut_by_levels_equals_the_whole_pass(us):
    """us_hashgrid_bwd_joint_part: record pass and accumulate pass of level ranges, in any order that keeps a level's record pass ahead of its
    accumulate pass, give the whole pass's gradient tables (unsplit bins: to the order of the f64 sums inside a bin)"""
    import ctypes
    from unislam_amd import _lib as L
    torch.manual_seed(3)
    n = 50000
    ea, eb = us.HashGridEncoding(3, enc_cfg(14, 816)).to(DEV), us.HashGridEncoding(3, enc_cfg(17, 816)).to(DEV)
    x = torch.rand(n, 3, device=DEV)
    dya, dyb = torch.randn(16, n, 2, device=DEV), torch.randn(16, n, 2, device=DEV)
    lib, P = L.lib(), L.ptr
    da, db = ctypes.byref(ea.desc), ctypes.byref(eb.desc)
    nb = int(lib.us_hashgrid_joint_workspace_bytes(da, db, n))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    fa, fb = torch.empty(16 * n * 2, device=DEV), torch.empty(16 * n * 2, device=DEV)
    st = L.stream()
    L.check(lib.us_hashgrid_fwd_joint(da, db, P(ea.params.detach()), P(eb.params.detach()), P(x), n, P(fa), P(fb), 3, P(ws), nb, st), "fwd")
    base = 3 | L.US_GRID_BWD_OVERWRITE | L.US_GRID_BWD_DETERMINISTIC
    ga, gb = torch.empty(ea.desc.n_params, device=DEV), torch.empty(eb.desc.n_params, device=DEV)
    L.check(lib.us_hashgrid_joint_scan(da, db, n, P(ga), P(gb), base, P(ws), nb, st), "scan")
    flags = base | L.US_GRID_BWD_COUNTED | L.US_GRID_BWD_SCANNED
    L.check(lib.us_hashgrid_bwd_joint(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), flags, P(ws), nb, st), "whole")
    ra, rb = ga.clone(), gb.clone()
    for order in (((0, 7, 1), (7, 16, 1), (7, 16, 2), (0, 7, 2)), ((0, 3, 3), (3, 11, 1), (11, 16, 3), (3, 11, 2))):
        ga.fill_(float("nan")); gb.fill_(float("nan"))
        for lo, hi, what in order:
            L.check(lib.us_hashgrid_bwd_joint_part(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), flags, P(ws), nb, lo, hi, what, st), "part")
        for got, want in ((ga, ra), (gb, rb)):
            assert torch.isfinite(got).all()
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert lib.us_hashgrid_bwd_joint_part(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), base, P(ws), nb, 0, 8, 1, st) == L.US_ERR_CONFIG   # no counts
    assert lib.us_hashgrid_bwd_joint_part(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), flags, P(ws), nb, 5, 5, 1, st) == -2               # empty range


@pytest.mark.parametrize("joint_opt", [False, True])
def test_adam_in_the_accumulate_sweep_equals_the_separate_pass(us, joint_opt):
    """MapStep.fuse_adam: the tables' optimiser step applied by the accumulate pass's sweep (us_hashgrid_bwd_joint_adam: the workgroup that
    owns an entry reads p, m, v and writes them back) against the separate optimiser launch (us_adam_step_segments_dev) applied to the
    SAME gradient -- the one the fused pass leaves with keep_table_grad -- from the same state: parameters and both moments of both tables
    BIT FOR BIT, three iterations running (incl. entries of bins nothing lands in: the far levels of a small batch); and a whole fused
    run lands where a run with the separate pass lands (to the rounding of the f64 sums, which differ from run to run in a few entries)."""
    import ctypes
    from unislam_amd import _lib as L
    b, P, n_per = 6, 500, 100
    c2ws, depths, colors, dirs = _window(b, P, 31)
    g = torch.Generator().manual_seed(4)
    draws = [(torch.randint(P, (b, n_per), generator=g).to(DEV), torch.rand(b * n_per, 40, generator=g).to(DEV)) for _ in range(4)]

    def build(fuse):
        torch.manual_seed(0)
        dec = us.Decoders(dict(_wcfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(16)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=b * n_per)
        step.fuse_adam, step.keep_table_grad = fuse, fuse
        return step, us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=joint_opt, cam_lr=1e-3, has_zero_depth=False)

    step, win = build(True)
    segs = [(step.o_tab_s, step.es.desc.n_params, LR["sdf_grid"]), (step.o_tab_c, step.ec.desc.n_params, LR["color_grid"])]
    I64, DBL = ctypes.c_int64 * 2, ctypes.c_double * 2
    for idx, tr in draws[:3]:
        p0, m0, v0, sd0 = step.flat.clone(), step.m.clone(), step.v.clone(), step.step_dev.clone()
        win.iterate(idx, None, t_rand=tr)
        gcopy = step.grad.clone()                                 # (the fused pass wrote the table segments: keep_table_grad)
        L.check(L.lib().us_adam_step_segments_dev(L.ptr(p0), L.ptr(gcopy), L.ptr(m0), L.ptr(v0), 2, I64(*[s_[0] for s_ in segs]), I64(*[s_[1] for s_ in segs]),
                                                  DBL(*[s_[2] for s_ in segs]), 0.9, 0.999, 1e-8, L.ptr(sd0), 0, L.stream()), "adam")
        assert torch.equal(sd0, step.step_dev)                    # the same step count and bias corrections
        for got, want, name in ((step.flat, p0, "parameters"), (step.m, m0, "first moments"), (step.v, v0, "second moments")):
            assert torch.equal(got[step.o_tab_s:], want[step.o_tab_s:]), name
        assert float((step.m[step.o_tab_s:] != 0).float().mean()) > 0.01
    # a whole run either way
    outs = []
    for fuse in (False, True):
        step, win = build(fuse)
        losses = [float(win.iterate(idx, None, t_rand=tr)) for idx, tr in draws]
        outs.append((step.flat.clone(), win.poses.clone(), losses))
    np.testing.assert_allclose(outs[1][2], outs[0][2], rtol=1e-5)
    d = (outs[0][0] - outs[1][0]).abs()
    assert float((d > 1e-6).float().mean()) < 1e-4 and float(d.max()) < 2e-3, (float((d > 1e-6).float().mean()), float(d.max()))
    assert torch.allclose(outs[0][1], outs[1][1], rtol=0, atol=1e-6)
