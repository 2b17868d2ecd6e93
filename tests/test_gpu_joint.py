"""
GPU: the two-grid kernels (csrc/hashgrid_joint.hip: us_hashgrid_fwd_joint / us_hashgrid_bwd_joint) against the one-grid kernels
and the CPU oracle.  Bars: features bit-identical to us_hashgrid_fwd; table gradients equal to us_hashgrid_bwd_binned up to the
rounding of one f64 -> f32 conversion (the sums are formed in double in both; 1e-6 relative) and to the oracle at 1e-4.
Pairs: room0 (log2T 16 / 19: dense-dense, hashed-dense and hashed-hashed levels), ScanNet / TUM (16 / 16) and the default 19 / 19.
"""
import ctypes

import numpy as np
import pytest
import torch

import unislam_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _cfg(log2T, res=816):
    return {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2T, "base_resolution": 16,
            "per_level_scale": O.per_level_scale(res)}


def _ray_points(n, seed, spread=0.8):
    """ray-like order: runs of consecutive samples inside one cell, as the binning's run-combining expects; a few coordinates outside [0, 1]"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    R = n // 64 + 1
    o = torch.rand((R, 1, 3), device=DEV, generator=g) * spread + (1 - spread) / 2
    dirs = torch.randn((R, 1, 3), device=DEV, generator=g) * 0.25
    t = torch.linspace(0, 1, 64, device=DEV).reshape(1, 64, 1)
    return (o + dirs * t).reshape(-1, 3)[:n].contiguous(), g


def _pair(us, l2a, l2b, res, g):
    ea, eb = us.HashGridEncoding(3, _cfg(l2a, res)).to(DEV), us.HashGridEncoding(3, _cfg(l2b, res)).to(DEV)
    with torch.no_grad():
        ea.params.copy_(torch.randn(ea.params.shape, device=DEV, generator=g) * 0.2)
        eb.params.copy_(torch.randn(eb.params.shape, device=DEV, generator=g) * 0.2)
    return ea, eb


@pytest.fixture(scope="module")
def us():
    import unislam_amd
    assert torch.cuda.is_available()
    return unislam_amd


@pytest.mark.parametrize("l2a,l2b,res,n", [(16, 19, 816, 70001), (16, 16, 456, 40000), (19, 19, 816, 9000), (16, 19, 816, 1), (14, 15, 816, 2047),
                                           (19, 16, 816, 5000)])
def test_joint_forward_is_the_single_grid_forward(us, l2a, l2b, res, n):
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    x, g = _ray_points(n, 100 + n)
    ea, eb = _pair(us, l2a, l2b, res, g)
    da, db = ctypes.byref(ea.desc), ctypes.byref(eb.desc)
    assert lib.us_hashgrid_joint_supported(da, db, n) == 1
    for flags in (3, 1, 2):                                       # clamp + level-major, clamp + row-major, level-major without clamp
        shape = (16, n, 2) if flags & 2 else (n, 32)
        xin = x if flags & 1 else x.clamp(0, 1)
        ra, rb, oa, ob = (torch.empty(shape, device=DEV) for _ in range(4))
        L.check(lib.us_hashgrid_fwd(da, P(ea.params.detach()), P(xin), n, P(ra), None, flags, st), "fwd a")
        L.check(lib.us_hashgrid_fwd(db, P(eb.params.detach()), P(xin), n, P(rb), None, flags, st), "fwd b")
        L.check(lib.us_hashgrid_fwd_joint(da, db, P(ea.params.detach()), P(eb.params.detach()), P(xin), n, P(oa), P(ob), flags, None, 0, st), "fwd joint")
        assert torch.equal(oa, ra) and torch.equal(ob, rb)
        nbytes = int(lib.us_hashgrid_joint_workspace_bytes(da, db, n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        oa.zero_(); ob.zero_()
        L.check(lib.us_hashgrid_fwd_joint(da, db, P(ea.params.detach()), P(eb.params.detach()), P(xin), n, P(oa), P(ob), flags, P(ws), nbytes, st), "fwd joint counted")
        assert torch.equal(oa, ra) and torch.equal(ob, rb)


def _single_grad(us, enc, x, dy, n):
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    d = ctypes.byref(enc.desc)
    nbytes = int(lib.us_hashgrid_bwd_workspace_bytes(d, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    g = torch.full((enc.desc.n_params,), 5.0, device=DEV)
    L.check(lib.us_hashgrid_bwd_binned(d, P(x), P(dy), n, P(g), 3 | L.US_GRID_BWD_OVERWRITE, P(ws), nbytes, st), "bwd single")
    return g


@pytest.mark.parametrize("l2a,l2b,res,n", [(16, 19, 816, 70001), (16, 16, 456, 40000), (19, 19, 816, 9000), (16, 19, 816, 1), (14, 15, 816, 2047),
                                           (19, 16, 816, 5000), (16, 19, 816, 262144)])
def test_joint_backward_equals_the_single_grid_backward(us, l2a, l2b, res, n):
    """counted (the forward left the counts) and uncounted, OVERWRITE and accumulate; zero-gradient samples; n not a multiple of 1024"""
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    x, g = _ray_points(n, 200 + n)
    ea, eb = _pair(us, l2a, l2b, res, g)
    da, db = ctypes.byref(ea.desc), ctypes.byref(eb.desc)
    dya, dyb = torch.randn((16, n, 2), device=DEV, generator=g), torch.randn((16, n, 2), device=DEV, generator=g)
    dya[:, ::9] = 0.0; dyb[:, ::7] = 0.0
    ga_ref, gb_ref = _single_grad(us, ea, x, dya, n), _single_grad(us, eb, x, dyb, n)
    nbytes = int(lib.us_hashgrid_joint_workspace_bytes(da, db, n))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    tol = lambda a, b: torch.allclose(a, b, rtol=1e-6, atol=1e-7 * float(b.abs().max()))
    # uncounted, OVERWRITE on garbage
    ga, gb = torch.full_like(ga_ref, 9.0), torch.full_like(gb_ref, -4.0)
    L.check(lib.us_hashgrid_bwd_joint(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), 3 | L.US_GRID_BWD_OVERWRITE, P(ws), nbytes, st), "bwd joint")
    assert tol(ga, ga_ref), float((ga - ga_ref).abs().max())
    assert tol(gb, gb_ref), float((gb - gb_ref).abs().max())
    # counted by the joint forward, accumulate onto a base
    oa, ob = torch.empty((16, n, 2), device=DEV), torch.empty((16, n, 2), device=DEV)
    L.check(lib.us_hashgrid_fwd_joint(da, db, P(ea.params.detach()), P(eb.params.detach()), P(x), n, P(oa), P(ob), 3, P(ws), nbytes, st), "fwd joint")
    base_a, base_b = torch.randn_like(ga_ref), torch.randn_like(gb_ref)
    ga, gb = base_a.clone(), base_b.clone()
    L.check(lib.us_hashgrid_bwd_joint(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), 3 | L.US_GRID_BWD_COUNTED, P(ws), nbytes, st), "bwd joint counted")
    assert torch.allclose(ga - base_a, ga_ref, rtol=1e-5, atol=1e-6 * float(ga_ref.abs().max()) + 1e-6)
    assert torch.allclose(gb - base_b, gb_ref, rtol=1e-5, atol=1e-6 * float(gb_ref.abs().max()) + 1e-6)
    if n <= 70001:                                                # and against the CPU oracle
        for enc, dy, gg in ((ea, dya, ga_ref), (eb, dyb, gb_ref)):
            d = O.make_grid_desc(16, 2, enc.desc.log2_hashmap_size, 16, O.per_level_scale(res))
            rows = dy.permute(1, 0, 2).reshape(n, 32).cpu().numpy()
            gp = O.hashgrid_bwd_params(d, x.clamp(0, 1).cpu().numpy(), rows)
            np.testing.assert_allclose(gg.cpu().numpy(), gp, rtol=1e-4, atol=1e-5 * np.abs(gp).max())


def test_joint_backward_hot_bins_and_empty(us):
    """60000 points inside one coarse cell: bins of more than ACC_CHUNK records are split over several accumulate workgroups (float
    atomics into entries the scan pass cleared); n == 0 with OVERWRITE clears both gradients."""
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    g = torch.Generator(device=DEV).manual_seed(5)
    n = 60000
    x = (0.41 + 0.02 * torch.rand((n, 3), device=DEV, generator=g)).contiguous()
    x[:5000] = torch.rand((5000, 3), device=DEV, generator=g)
    ea, eb = _pair(us, 16, 19, 816, g)
    da, db = ctypes.byref(ea.desc), ctypes.byref(eb.desc)
    dya, dyb = torch.randn((16, n, 2), device=DEV, generator=g), torch.randn((16, n, 2), device=DEV, generator=g)
    ga_ref, gb_ref = _single_grad(us, ea, x, dya, n), _single_grad(us, eb, x, dyb, n)
    nbytes = int(lib.us_hashgrid_joint_workspace_bytes(da, db, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    ga, gb = torch.full_like(ga_ref, 9.0), torch.full_like(gb_ref, -4.0)
    L.check(lib.us_hashgrid_bwd_joint(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), 3 | L.US_GRID_BWD_OVERWRITE, P(ws), nbytes, st), "bwd joint")
    assert torch.allclose(ga, ga_ref, rtol=1e-4, atol=2e-6 * float(ga_ref.abs().max()))
    assert torch.allclose(gb, gb_ref, rtol=1e-4, atol=2e-6 * float(gb_ref.abs().max()))
    L.check(lib.us_hashgrid_bwd_joint(da, db, None, None, None, 0, P(ga), P(gb), 3 | L.US_GRID_BWD_OVERWRITE, None, 0, st), "bwd joint n=0")
    assert float(ga.abs().max()) == 0.0 and float(gb.abs().max()) == 0.0


def test_joint_rejects_grids_of_different_geometry(us):
    from unislam_amd import _lib as L
    lib = L.lib()
    ea, eb = us.HashGridEncoding(3, _cfg(16, 816)), us.HashGridEncoding(3, _cfg(16, 456))
    assert lib.us_hashgrid_joint_supported(ctypes.byref(ea.desc), ctypes.byref(eb.desc), 1000) == 0
    assert lib.us_hashgrid_joint_workspace_bytes(ctypes.byref(ea.desc), ctypes.byref(eb.desc), 1000) == 0
    rc = lib.us_hashgrid_bwd_joint(ctypes.byref(ea.desc), ctypes.byref(eb.desc), None, None, None, 1000, None, None, 3, None, 0, None)
    assert rc == L.US_ERR_CONFIG


@pytest.mark.parametrize("joint", [True, False])
def test_deterministic_table_gradient_repeats_bit_for_bit(us, joint):
    """US_GRID_BWD_DETERMINISTIC: no bin is split, no float atomic takes part -- two runs on a batch that concentrates 55000 points in
    one coarse cell give bit-identical gradients (the default mode sums the chunks of such bins with float atomics, in arrival order),
    and they equal the default mode's up to that order."""
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    g = torch.Generator(device=DEV).manual_seed(6)
    n = 60000
    x = (0.41 + 0.02 * torch.rand((n, 3), device=DEV, generator=g)).contiguous()
    x[:5000] = torch.rand((5000, 3), device=DEV, generator=g)
    ea, eb = _pair(us, 16, 19, 816, g)
    da, db = ctypes.byref(ea.desc), ctypes.byref(eb.desc)
    dya, dyb = torch.randn((16, n, 2), device=DEV, generator=g), torch.randn((16, n, 2), device=DEV, generator=g)
    outs = []
    for flags in (L.US_GRID_BWD_DETERMINISTIC, L.US_GRID_BWD_DETERMINISTIC, 0):
        ga, gb = torch.full((ea.desc.n_params,), 3.0, device=DEV), torch.full((eb.desc.n_params,), 3.0, device=DEV)
        if joint:
            nbytes = int(lib.us_hashgrid_joint_workspace_bytes(da, db, n))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
            L.check(lib.us_hashgrid_bwd_joint(da, db, P(x), P(dya), P(dyb), n, P(ga), P(gb), 3 | L.US_GRID_BWD_OVERWRITE | flags, P(ws), nbytes, st), "bwd joint")
        else:
            for d, dy, gg in ((da, dya, ga), (db, dyb, gb)):
                nbytes = int(lib.us_hashgrid_bwd_workspace_bytes(d, n))
                ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
                L.check(lib.us_hashgrid_bwd_binned(d, P(x), P(dy), n, P(gg), 3 | L.US_GRID_BWD_OVERWRITE | flags, P(ws), nbytes, st), "bwd")
        outs.append((ga, gb))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in range(2):
        assert torch.allclose(outs[0][k], outs[2][k], rtol=1e-4, atol=2e-6 * float(outs[2][k].abs().max()))


@pytest.mark.parametrize("n", [70001, 4096 * 64, 31])
def test_pre_split_feature_planes_give_the_same_bits(n):
    """US_GRID_FEAT_SPLIT_BF16 + US_MLP_IN_SPLIT_BF16 (r6): the joint encoder writes the features as the split-bf16 decoders' hi / lo operand
    pairs, the decoders load them instead of splitting float planes -- outputs, input gradients and parameter gradients of the decoder pair
    BIT FOR BIT those of float planes (forward, backward with and without parameter gradients), and the planes themselves are the split of
    the float planes value for value."""
    import ctypes
    import unislam_amd as us
    from unislam_amd import _lib as L
    torch.manual_seed(11)
    lib, P, st = L.lib(), L.ptr, L.stream()
    ea, eb = us.HashGridEncoding(3, _cfg(16)).to(DEV), us.HashGridEncoding(3, _cfg(19)).to(DEV)
    with torch.no_grad():
        ea.params.copy_(torch.randn_like(ea.params) * 0.1); eb.params.copy_(torch.randn_like(eb.params) * 0.1)
    x = torch.rand(n, 3, device=DEV)
    da, db = ctypes.byref(ea.desc), ctypes.byref(eb.desc)
    ds = us.make_mlp_desc(32, 32, 2, 1, "tanh", True, "bf16"); dc = us.make_mlp_desc(32, 32, 2, 3, "sigmoid", True, "bf16")
    ps = torch.randn(us.network.mlp_n_params(ds), device=DEV) * 0.3; pc = torch.randn(us.network.mlp_n_params(dc), device=DEV) * 0.3
    A, B = ctypes.byref(ds), ctypes.byref(dc)
    d_raw = torch.randn(n, 4, device=DEV)
    off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
    wsb = int(lib.us_mlp_bwd_workspace_bytes(A))
    outs = []
    for split in (0, 1):
        fa, fb = torch.empty(16, n, 2, device=DEV), torch.empty(16, n, 2, device=DEV)
        L.check(lib.us_hashgrid_fwd_joint(da, db, P(ea.params.detach()), P(eb.params.detach()), P(x), n, P(fa), P(fb),
                                          3 | (L.US_GRID_FEAT_SPLIT_BF16 if split else 0), None, 0, st), "fwd")
        mf = 1 | (L.US_MLP_IN_SPLIT_BF16 if split else 0)
        raw = torch.empty(n, 4, device=DEV)
        L.check(lib.us_mlp_fwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), n, off(raw, 3), 4, P(raw), 4, mf, st), "mlp fwd")
        dfa, dfb = torch.empty(16, n, 2, device=DEV), torch.empty(16, n, 2, device=DEV)
        gs, gc = torch.zeros_like(ps), torch.zeros_like(pc)
        wa, wb = torch.empty(wsb, dtype=torch.uint8, device=DEV), torch.empty(wsb, dtype=torch.uint8, device=DEV)
        L.check(lib.us_mlp_bwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), off(raw, 3), 4, P(raw), 4, off(d_raw, 3), 4, P(d_raw), 4, n, P(dfa), P(dfb),
                                    P(gs), P(gc), mf, P(wa), P(wb), wsb, st), "mlp bwd")
        dfa2, dfb2 = torch.empty(16, n, 2, device=DEV), torch.empty(16, n, 2, device=DEV)
        L.check(lib.us_mlp_bwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), off(raw, 3), 4, P(raw), 4, off(d_raw, 3), 4, P(d_raw), 4, n, P(dfa2), P(dfb2),
                                    None, None, mf, None, None, 0, st), "mlp bwd (inputs only)")
        outs.append((fa, fb, raw, dfa, dfb, gs, gc, dfa2, dfb2))
    f0, f1 = outs
    for k, name in ((2, "raw"), (3, "dL/dfeatures (sdf)"), (4, "dL/dfeatures (colour)"), (5, "sdf decoder gradient"), (6, "colour decoder gradient"),
                    (7, "dL/dfeatures, lean kernel (sdf)"), (8, "dL/dfeatures, lean kernel (colour)")):
        assert torch.equal(f0[k], f1[k]), name
    # the planes: hi = bf16(f), lo = bf16(f - hi), two features per dword pair
    for k in (0, 1):
        f = f0[k].reshape(-1, 2)
        hi = f.bfloat16()
        lo = (f - hi.float()).bfloat16()
        w = f1[k].reshape(-1, 2).view(torch.int32)
        bits = lambda t: t.view(torch.int16).to(torch.int32) & 0xFFFF
        assert torch.equal(w[:, 0], bits(hi[:, 0]) | (bits(hi[:, 1]) << 16)) and torch.equal(w[:, 1], bits(lo[:, 0]) | (bits(lo[:, 1]) << 16))
    # the flag is refused where it does not apply
    dp = us.make_mlp_desc(32, 32, 2, 1, "tanh", True, "bf16_plain")
    raw = torch.empty(n, 4, device=DEV)
    assert lib.us_mlp_fwd(ctypes.byref(dp), P(ps), P(f1[0]), n, off(raw, 3), 4, 1 | L.US_MLP_IN_SPLIT_BF16, st) == L.US_ERR_CONFIG
    assert lib.us_mlp_fwd(A, P(ps), P(f1[0]), n, off(raw, 3), 4, L.US_MLP_IN_SPLIT_BF16, st) == L.US_ERR_CONFIG            # row-major planes
    assert lib.us_hashgrid_fwd_joint(da, db, P(ea.params.detach()), P(eb.params.detach()), P(x), n, P(f1[0]), P(f1[1]), 1 | L.US_GRID_FEAT_SPLIT_BF16,
                                     None, 0, st) == L.US_ERR_CONFIG
