"""
GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI
(libunislam_hip.so via unislam_amd), against the CPU oracle and the committed golden fixtures.
Tolerances: hash indices and z_vals bit-exact; fp32 values within the rtol written at each check
(north_star: 1e-3 relative on rendered RGB/depth; the fp32 kernels are held much tighter).
"""
import numpy as np
import pytest
import torch

import unislam_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"
PLS816 = O.per_level_scale(816)
BOUND = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])


@pytest.fixture(scope="module")
def us():
    import unislam_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return unislam_amd


def enc_cfg(log2T, res=816, L=16, F=2):
    return {"otype": "HashGrid", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": log2T,
            "base_resolution": 16, "per_level_scale": O.per_level_scale(res)}


def close(a, b, rtol, atol):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ---------------------------------------------------------------------------------------------- hash grid
@pytest.mark.parametrize("log2T,n", [(16, 4099), (19, 1000), (10, 1), (12, 64)])
def test_hashgrid_indices_bit_exact_and_features(us, log2T, n):
    rng = np.random.default_rng(log2T + n)
    x = rng.random((n, 3), dtype=np.float32)
    x[0] = 0.0
    if n > 2:
        x[1] = 1.0; x[2] = [1.0, 0.0, 0.5]
    enc = us.HashGridEncoding(3, enc_cfg(log2T)).to(DEV)
    d = O.make_grid_desc(16, 2, log2T, 16, PLS816)
    assert enc.desc.n_params == d.n_params
    p = (rng.random(d.n_params, dtype=np.float32) * 2 - 1)
    with torch.no_grad():
        enc.params.copy_(T(p))
    xg = T(x).to(DEV)
    idx = us.grid_indices(enc.desc, xg).cpu().numpy().astype(np.uint32)
    assert np.array_equal(idx, O.hashgrid_indices(d, x))                       # bit-exact indices
    out = enc(xg).detach().cpu().numpy()
    ref, _ = O.hashgrid_fwd(d, p, x)
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-7)                  # same fmaf chain


def test_hashgrid_clamp_and_out_of_range(us):
    rng = np.random.default_rng(3)
    x = (rng.random((500, 3), dtype=np.float32) * 1.6 - 0.3)                    # some coordinates outside [0,1]
    enc = us.HashGridEncoding(3, enc_cfg(14)).to(DEV)
    d = O.make_grid_desc(16, 2, 14, 16, PLS816)
    p = rng.standard_normal(d.n_params).astype(np.float32)
    with torch.no_grad():
        enc.params.copy_(T(p))
    xg = T(x).to(DEV).requires_grad_(True)
    out = enc(xg, clamp=True)
    xc = np.clip(x, 0, 1)
    ref, dydx = O.hashgrid_fwd(d, p, xc, True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    assert np.array_equal(us.grid_indices(enc.desc, xg.detach(), clamp=True).cpu().numpy().astype(np.uint32),
                          O.hashgrid_indices(d, xc))
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(T(dy).to(DEV))
    gx = O.hashgrid_bwd_input(dy, dydx) * ((x >= 0) & (x <= 1))                 # torch.clamp backward mask
    np.testing.assert_allclose(xg.grad.cpu().numpy(), gx, rtol=2e-4, atol=2e-3)
    # without the clamp flag the raw coordinates are hashed as they are (tcnn behaviour)
    assert np.array_equal(us.grid_indices(enc.desc, xg.detach()).cpu().numpy().astype(np.uint32), O.hashgrid_indices(d, x))
    # cells of slightly negative coordinates: the uint32 entry sum of a dense level reaches 0xFFFFFFFF (x cell -1, y = z = 0),
    # which a 16-byte "x, x+1" pair gather must treat as a wrap-around (regression: out-of-bounds read)
    xs = np.array([[-1.0 / 15, 0.01, 0.01], [-0.06, 0.0, 0.02], [-0.05, -0.01, 0.0], [-0.04, 0.02, -0.02], [1.02, 1.0, 1.0]], np.float32)
    xs = np.concatenate([xs, -rng.random((200, 3), dtype=np.float32) * 0.1 + np.array([[0.0, 0.04, 0.04]], np.float32)])
    o2 = enc(T(xs).to(DEV))
    r2, _ = O.hashgrid_fwd(d, p, xs)
    np.testing.assert_allclose(o2.detach().cpu().numpy(), r2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, -1])
@pytest.mark.parametrize("log2T,n", [(16, 20000), (19, 3000)])
def test_hashgrid_backward(us, mode, log2T, n):
    rng = np.random.default_rng(7 + log2T)
    x = rng.random((n, 3), dtype=np.float32)
    enc = us.HashGridEncoding(3, enc_cfg(log2T)).to(DEV)
    enc.bwd_mode = mode
    d = O.make_grid_desc(16, 2, log2T, 16, PLS816)
    p = rng.standard_normal(d.n_params).astype(np.float32) * 0.1
    with torch.no_grad():
        enc.params.copy_(T(p))
    xg = T(x).to(DEV).requires_grad_(True)
    dy = rng.standard_normal((n, 32)).astype(np.float32)
    dy[::7] = 0.0                                                              # rows the kernels may skip
    enc(xg).backward(T(dy).to(DEV))
    gp = O.hashgrid_bwd_params(d, x, dy)
    scale = np.abs(gp).max()
    np.testing.assert_allclose(enc.params.grad.cpu().numpy(), gp, rtol=1e-4, atol=1e-5 * scale)
    _, dydx = O.hashgrid_fwd(d, p, x, True)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), O.hashgrid_bwd_input(dy, dydx), rtol=2e-4, atol=1e-3)


def test_binned_backward_overwrite_and_split_bins(us):
    """us_hashgrid_bwd_binned: US_GRID_BWD_OVERWRITE writes every entry (grad buffer pre-filled with garbage), the default
    adds to what is there; 60000 points inside one coarse cell make bins of > ACC_CHUNK records, which several workgroups
    accumulate (float atomics into entries cleared by the scan pass)."""
    import ctypes
    from unislam_amd import _lib as L
    rng = np.random.default_rng(11)
    n = 60000
    x = (0.41 + 0.02 * rng.random((n, 3))).astype(np.float32)
    x[:5000] = rng.random((5000, 3), dtype=np.float32)
    dy = rng.standard_normal((n, 32)).astype(np.float32)
    for log2T in (14, 19):
        d = O.make_grid_desc(16, 2, log2T, 16, PLS816)
        enc = us.HashGridEncoding(3, enc_cfg(log2T)).to(DEV)
        gp = O.hashgrid_bwd_params(d, x, dy)
        scale = np.abs(gp).max()
        xd, dyd = T(x).to(DEV), T(dy).to(DEV)
        nbytes = int(L.lib().us_hashgrid_bwd_workspace_bytes(ctypes.byref(enc.desc), n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        g = torch.full((d.n_params,), 123.0, device=DEV)
        L.check(L.lib().us_hashgrid_bwd_binned(ctypes.byref(enc.desc), L.ptr(xd), L.ptr(dyd), n, L.ptr(g), L.US_GRID_BWD_OVERWRITE,
                                               L.ptr(ws), nbytes, L.stream()), "binned")
        np.testing.assert_allclose(g.cpu().numpy(), gp, rtol=1e-4, atol=2e-6 * scale)
        base = torch.randn(d.n_params, device=DEV)
        g2 = base.clone()
        L.check(L.lib().us_hashgrid_bwd_binned(ctypes.byref(enc.desc), L.ptr(xd), L.ptr(dyd), n, L.ptr(g2), 0, L.ptr(ws), nbytes,
                                               L.stream()), "binned")
        np.testing.assert_allclose((g2 - base).cpu().numpy(), gp, rtol=1e-4, atol=2e-6 * scale + 1e-6)
        # n == 0 with OVERWRITE clears the gradient
        L.check(L.lib().us_hashgrid_bwd_binned(ctypes.byref(enc.desc), None, None, 0, L.ptr(g), L.US_GRID_BWD_OVERWRITE, None, 0, L.stream()), "binned")
        assert float(g.abs().max()) == 0.0


def test_level_major_layout_matches_row_major(us):
    """the fused path's [L][N][F] planes (US_GRID_LEVEL_MAJOR / US_MLP_LEVEL_MAJOR) against the torch-view layout"""
    import ctypes
    from unislam_amd import _lib as L
    g = torch.Generator(device=DEV).manual_seed(1)
    n = 3001
    x = torch.rand((n, 3), device=DEV, generator=g)
    enc = us.HashGridEncoding(3, enc_cfg(15)).to(DEV)
    with torch.no_grad():
        enc.params.copy_(torch.randn(enc.params.shape, device=DEV, generator=g))
    rm = enc(x).detach()
    lm = torch.empty((16, n, 2), device=DEV)
    L.check(L.lib().us_hashgrid_fwd(ctypes.byref(enc.desc), L.ptr(enc.params.detach()), L.ptr(x), n, L.ptr(lm), None, 2, L.stream()), "fwd")
    assert torch.equal(lm.permute(1, 0, 2).reshape(n, 32), rm)
    dy = torch.randn((n, 32), device=DEV, generator=g)
    dy_lm = dy.view(n, 16, 2).permute(1, 0, 2).contiguous()
    for mode in (0, 1):
        ga, gb = torch.zeros_like(enc.params), torch.zeros_like(enc.params)
        L.check(L.lib().us_hashgrid_bwd_params(ctypes.byref(enc.desc), L.ptr(x), L.ptr(dy), n, L.ptr(ga), mode, 0, L.stream()), "bwd")
        L.check(L.lib().us_hashgrid_bwd_params(ctypes.byref(enc.desc), L.ptr(x), L.ptr(dy_lm), n, L.ptr(gb), mode, 2, L.stream()), "bwd")
        assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-5 * ga.abs().max().item())
    # MLP on level-major input / input-gradient
    desc = us.make_mlp_desc(32, 32, 2, 3, "sigmoid", True)
    p = torch.randn(us.network.mlp_n_params(desc), device=DEV, generator=g) * 0.3
    y_rm, y_lm = torch.empty((n, 3), device=DEV), torch.empty((n, 3), device=DEV)
    L.check(L.lib().us_mlp_fwd(ctypes.byref(desc), L.ptr(p), L.ptr(rm), n, L.ptr(y_rm), 3, 0, L.stream()), "mlp")
    L.check(L.lib().us_mlp_fwd(ctypes.byref(desc), L.ptr(p), L.ptr(lm), n, L.ptr(y_lm), 3, 1, L.stream()), "mlp")
    assert torch.equal(y_rm, y_lm)
    dyo = torch.randn((n, 3), device=DEV, generator=g)
    dx_rm, dx_lm = torch.empty((n, 32), device=DEV), torch.empty((16, n, 2), device=DEV)
    g1, g2 = torch.zeros_like(p), torch.zeros_like(p)
    L.check(L.lib().us_mlp_bwd(ctypes.byref(desc), L.ptr(p), L.ptr(rm), L.ptr(y_rm), 3, L.ptr(dyo), 3, n, L.ptr(dx_rm), L.ptr(g1), 0, None, 0, L.stream()), "mlpb")
    L.check(L.lib().us_mlp_bwd(ctypes.byref(desc), L.ptr(p), L.ptr(lm), L.ptr(y_lm), 3, L.ptr(dyo), 3, n, L.ptr(dx_lm), L.ptr(g2), 1, None, 0, L.stream()), "mlpb")
    assert torch.equal(dx_lm.permute(1, 0, 2).reshape(n, 32), dx_rm)
    assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-5 * g1.abs().max().item())


def test_hashgrid_empty_and_features_1_4(us):
    enc = us.HashGridEncoding(3, enc_cfg(12)).to(DEV)
    assert enc(torch.empty(0, 3, device=DEV)).shape == (0, 32)
    rng = np.random.default_rng(5)
    x = rng.random((333, 3), dtype=np.float32)
    for F in (1, 4):
        e = us.HashGridEncoding(3, enc_cfg(11, res=300, L=8, F=F)).to(DEV)
        d = O.make_grid_desc(8, F, 11, 16, O.per_level_scale(300))
        p = rng.standard_normal(d.n_params).astype(np.float32)
        with torch.no_grad():
            e.params.copy_(T(p))
        xg = T(x).to(DEV)
        out = e(xg)
        ref, _ = O.hashgrid_fwd(d, p, x)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
        dy = rng.standard_normal(ref.shape).astype(np.float32)
        out.backward(T(dy).to(DEV))
        gp = O.hashgrid_bwd_params(d, x, dy)
        np.testing.assert_allclose(e.params.grad.cpu().numpy(), gp, rtol=1e-4, atol=1e-5 * np.abs(gp).max())


def test_hashgrid_full_size_properties(us):
    """BASELINE cfg2 size (4096 rays x 64 samples, room0 tables): size-independent properties."""
    n = 4096 * 64
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.rand((n, 3), device=DEV, generator=g)
    for log2T in (16, 19):
        enc = us.HashGridEncoding(3, enc_cfg(log2T)).to(DEV)
        with torch.no_grad():
            enc.params.fill_(0.25)
            assert torch.allclose(enc(x), torch.full((n, 32), 0.25, device=DEV), rtol=1e-6)     # partition of unity
            p1 = torch.randn(enc.params.shape, device=DEV, generator=g); p2 = torch.randn(enc.params.shape, device=DEV, generator=g)
            enc.params.copy_(p1); o1 = enc(x)
            enc.params.copy_(p2); o2 = enc(x)
            enc.params.copy_(p1 + 2 * p2); o12 = enc(x)
            assert torch.allclose(o12, o1 + 2 * o2, rtol=1e-4, atol=1e-4)                       # linear in the table
        # adjoint identity <dy, fwd(q)> == <bwd(dy), q>, for both backward strategies, and mode 0 == mode 1
        dy = torch.randn((n, 32), device=DEV, generator=g)
        grads = []
        for mode in (0, 1, 3):
            enc.bwd_mode = mode
            enc.params.grad = None
            out = enc(x)
            out.backward(dy)
            lhs = (dy.double() * out.detach().double()).sum()
            rhs = (enc.params.grad.double() * enc.params.detach().double()).sum()
            assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-2
            grads.append(enc.params.grad.clone())
        for gk in grads[1:]:
            assert torch.allclose(grads[0], gk, rtol=1e-3, atol=1e-3 * grads[0].abs().max().item())


# ---------------------------------------------------------------------------------------------- MLP
@pytest.mark.parametrize("width,n_hidden,n_out,act,bias,n", [
    (16, 1, 1, "tanh", False, 1000),        # reference tcnn path, SDF   (decoders.py:50-59)
    (16, 1, 3, "sigmoid", False, 257),      # reference tcnn path, colour
    (16, 2, 1, "tanh", True, 4096),         # reference torch path       (decoders.py:74-84)
    (16, 2, 3, "sigmoid", True, 63),
    (32, 2, 3, "sigmoid", True, 5000),      # BASELINE "2x32"
    (32, 1, 1, "tanh", True, 130),
    (64, 2, 3, "none", True, 700),
    (64, 1, 1, "tanh", False, 33),
])
def test_mlp_forward_backward(us, width, n_hidden, n_out, act, bias, n):
    g = torch.Generator().manual_seed(width * 10 + n_hidden)
    desc = us.make_mlp_desc(32, width, n_hidden, n_out, act, bias)
    n_p = us.network.mlp_n_params(desc)
    shapes = [(width, 32)] + [(width, width)] * (n_hidden - 1) + [(16, width)]
    assert n_p == sum(a * b for a, b in shapes) + (n_hidden * width + 16 if bias else 0)
    params = (torch.rand(n_p, generator=g) * 2 - 1) * 0.4
    x = torch.randn(n, 32, generator=g)
    dy = torch.randn(n, n_out, generator=g)
    # fp32 torch reference of the same flat layout
    pr = params.clone().requires_grad_(True); xr = x.clone().requires_grad_(True)
    ws, o = [], 0
    for (a, b) in shapes:
        ws.append(pr[o:o + a * b].view(a, b)); o += a * b
    bs = None
    if bias:
        bs = []
        for (a, _) in shapes:
            bs.append(pr[o:o + a]); o += a
    ws[-1] = ws[-1][:n_out]
    if bs:
        bs[-1] = bs[-1][:n_out]
    yr = O.mlp_forward(xr, ws, bs, act)
    (yr * dy).sum().backward()
    pg = params.to(DEV).requires_grad_(True); xg = x.to(DEV).requires_grad_(True)
    y = us.fused_mlp(xg, pg, desc)
    close(y, yr, 2e-5, 2e-6)
    (y * dy.to(DEV)).sum().backward()
    close(xg.grad, xr.grad, 1e-4, 1e-5)
    gref = pr.grad.clone()
    scale = gref.abs().max().item()
    close(pg.grad, gref, 1e-4, 1e-5 * scale)


def _bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize("width,n_hidden,n_out,act,bias,n", [
    (16, 1, 1, "tanh", False, 1000), (16, 2, 3, "sigmoid", True, 4099), (32, 2, 1, "tanh", True, 8192 + 17),
    (32, 2, 3, "sigmoid", True, 70001), (32, 1, 3, "sigmoid", False, 333), (64, 2, 3, "sigmoid", True, 5000), (64, 1, 1, "none", True, 31)])
@pytest.mark.parametrize("prec", ["bf16", "bf16_plain", "f16"])
def test_mlp_bf16(us, width, n_hidden, n_out, act, bias, n, prec):
    """bf16 / f16 MFMA decoders against a torch emulation that rounds the MFMA operands to that type (tight) and against fp32.
    bf16_plain: every operand of every product rounded once.  bf16 (US_PREC_BF16, the default): the forward products carry split
    operands (hi + lo), so outputs and activations are fp32-accurate (1e-4) and only the gradient products see bf16 operands.
    f16 (US_PREC_F16): one product with f16 operands (2^-11) everywhere -- the reference's tcnn arithmetic, src/networks/decoders.py:50-70 --,
    the gradient chain scaled per chunk; the dL/dout handed in spans eight decades, as loss gradients do."""
    g = torch.Generator().manual_seed(width * 7 + n_hidden)
    desc = us.make_mlp_desc(32, width, n_hidden, n_out, act, bias, prec)
    assert desc.precision == {"bf16": 1, "bf16_plain": 2, "f16": 3}[prec]
    shapes = [(width, 32)] + [(width, width)] * (n_hidden - 1) + [(16, width)]
    n_p = us.network.mlp_n_params(desc)
    params = (torch.rand(n_p, generator=g) * 2 - 1) * 0.4
    x = torch.randn(n, 32, generator=g)
    dy = torch.randn(n, n_out, generator=g)
    if prec == "f16":                                            # f16's exponent range must not show: gradients from 1e-9 to 1e-1, features at 1e-4
        dy = dy * torch.pow(10.0, -1.0 - 8.0 * torch.rand(n, 1, generator=g))
        x[: n // 2] *= 1e-4
    ws, bs, o = [], [], 0
    for (a, b) in shapes:
        ws.append(params[o:o + a * b].view(a, b)); o += a * b
    for (a, _) in shapes:
        bs.append(params[o:o + a] if bias else torch.zeros(a)); o += a
    def run(rnd, rnd_fwd=None):
        rf = rnd if rnd_fwd is None else rnd_fwd                  # operand rounding of the forward products
        hs, h = [rf(x)], rf(x)
        for W, b in zip(ws[:-1], bs[:-1]):
            h = rf(torch.relu(h @ rf(W).T + b)); hs.append(h)
        ypre = (h @ rf(ws[-1]).T + bs[-1])[:, :n_out]
        y = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "none": lambda t: t}[act](ypre)
        dact = {"tanh": 1 - y * y, "sigmoid": y * (1 - y), "none": torch.ones_like(y)}[act]
        d = torch.zeros(n, 16); d[:, :n_out] = dy * dact
        gw, gb = [], []
        for i in range(len(ws) - 1, -1, -1):
            gw.append(rnd(d).T @ rnd(hs[i])); gb.append(d.sum(0))
            d = rnd(d) @ rnd(ws[i])
            if i > 0:
                d = d * (hs[i] > 0)
        gp = torch.cat([t.reshape(-1) for t in gw[::-1]] + ([t for t in gb[::-1]] if bias else []))
        return y, d, gp
    ident = lambda t: t
    def _hf(t):                                                  # 11 significant bits at any exponent (the kernel's scalings keep f16's range out)
        m, e = torch.frexp(t)
        return torch.ldexp(torch.round(m * 2048.0) / 2048.0, e)
    y_e, dx_e, gp_e = run(_bf) if prec == "bf16_plain" else run(_hf) if prec == "f16" else run(_bf, ident)
    y_f, dx_f, gp_f = run(ident)
    pg = params.to(DEV).requires_grad_(True); xg = x.to(DEV).requires_grad_(True)
    y = us.fused_mlp(xg, pg, desc)
    (y * dy.to(DEV)).sum().backward()
    # last-matrix rows >= n_out receive no gradient in either implementation
    def rel(a, b):
        return ((a.cpu() - b).norm() / (b.norm() + 1e-12)).item()
    gtol = 3e-3 if prec in ("bf16_plain", "f16") else 1e-2       # (the emulation of the mixed path does not model where its sums round)
    assert rel(y, y_e) < 2e-3 and rel(xg.grad, dx_e) < gtol and rel(pg.grad, gp_e) < gtol, (rel(y, y_e), rel(xg.grad, dx_e), rel(pg.grad, gp_e))
    if prec == "bf16":
        # split operands: the forward pass is fp32-accurate; the gradients carry one bf16 rounding per operand of their products
        assert rel(y, y_f) < 1e-4 and (y.cpu() - y_f).abs().max().item() < 2e-4, (rel(y, y_f), (y.cpu() - y_f).abs().max().item())
        assert rel(xg.grad, dx_f) < 1e-2 and rel(pg.grad, gp_f) < 1e-2, (rel(xg.grad, dx_f), rel(pg.grad, gp_f))
    elif prec == "f16":
        # 8 x finer operands than bf16_plain: outputs to ~1e-3; gradients: ReLU masks at rounding distance from 0 flip (half of the inputs
        # are scaled to 1e-4 here, so without biases every pre-activation of those rows sits near 0) -- the tight check is the emulation above
        assert rel(y, y_f) < 3e-3 and rel(xg.grad, dx_f) < 0.15 and rel(pg.grad, gp_f) < 0.15, (rel(y, y_f), rel(xg.grad, dx_f), rel(pg.grad, gp_f))
        # a point's input gradient is as good relative to ITS OWN size whatever the size (per-chunk scale): the rows with the smallest dL/dout
        small = dy.abs().amax(1) < 1e-6
        if int(small.sum()) > 8:
            assert rel(xg.grad[small.to(DEV)], dx_e[small]) < 1e-2, rel(xg.grad[small.to(DEV)], dx_e[small])
    else:
        # vs fp32: ReLU masks of neurons whose pre-activation is within bf16 rounding of 0 flip, so gradients differ by a few %
        assert rel(y, y_f) < 2e-2 and rel(xg.grad, dx_f) < 0.15 and rel(pg.grad, gp_f) < 0.08, (rel(y, y_f), rel(xg.grad, dx_f), rel(pg.grad, gp_f))
        assert (y.cpu() - y_e).abs().max().item() < 2e-2


def test_mlp_rejects_unsupported(us):
    with pytest.raises(us.UniSlamHipError):
        us.fused_mlp(torch.zeros(4, 32, device=DEV), torch.zeros(10000, device=DEV), us.make_mlp_desc(32, 48, 1, 1, "tanh", False))
    with pytest.raises(us.UniSlamHipError):
        us.fused_mlp(torch.zeros(4, 32), torch.zeros(768), us.make_mlp_desc(32, 16, 1, 1, "tanh", False))   # CPU tensor


# ---------------------------------------------------------------------------------------------- sampler / compositing / losses
@pytest.mark.parametrize("ns,ni", [(32, 8), (48, 8)])
def test_sample_z_bit_exact(us, golden, ns, ni):
    g = golden("g2_zsample")
    gt = T(g["gt_depth"]).to(DEV)
    tu, ts = torch.linspace(0., 1., ns).to(DEV), torch.linspace(0., 1., ni).to(DEV)
    z0 = us.sample_z(gt, float(g["truncation"]), tu, ts)
    assert np.array_equal(z0.cpu().numpy(), g[f"z_{ns}_{ni}_0"])
    z1 = us.sample_z(gt, float(g["truncation"]), tu, ts, T(g[f"trand_{ns}_{ni}"]).to(DEV))
    assert np.array_equal(z1.cpu().numpy(), g[f"z_{ns}_{ni}_1"])


def test_sample_z_bench_shapes(us):
    # 64 = 48+16 and 96 = 80+16 samples (BASELINE cfg2/cfg3), ragged ray count
    for ns, ni, R in [(48, 16, 4099), (80, 16, 1001)]:
        gt = torch.rand(R) * 3 + 0.3
        tr = torch.rand(R, ns + ni)
        ref = O.sample_z_with_depth(gt[:, None], 0.06, ns, ni, True, tr)
        z = us.sample_z(gt.to(DEV), 0.06, torch.linspace(0., 1., ns).to(DEV), torch.linspace(0., 1., ni).to(DEV), tr.to(DEV))
        assert np.array_equal(z.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("tag", ["b10", "b73"])
def test_composite_golden(us, golden, tag):
    from unislam_amd.renderer import _CompositeFn
    g = golden("g3_composite")
    raw = T(g[f"{tag}_raw"]).to(DEV).requires_grad_(True)
    beta = T(g[f"{tag}_beta"]).to(DEV).requires_grad_(True)
    z = T(g[f"{tag}_z"]).to(DEV)
    term, unc, depth, rgb, dunc, sdf = _CompositeFn.apply(raw, z, beta)
    assert torch.equal(sdf, raw[..., 3].detach())
    outs = dict(term=term, unc=unc, depth=depth, rgb=rgb, dunc=dunc)
    for k, v in outs.items():
        close(v, g[f"{tag}_{k}"], 2e-5, 2e-6)
    sum((T(g[f"{tag}_probe_{k}"]).to(DEV) * v).sum() for k, v in outs.items()).backward()
    close(raw.grad, g[f"{tag}_draw"], 2e-4, 2e-5)
    close(beta.grad, g[f"{tag}_dbeta"], 2e-4, 2e-4)


def test_composite_96_samples_vs_oracle(us):
    from unislam_amd.renderer import _CompositeFn
    g = torch.Generator().manual_seed(11)
    R, S = 301, 96
    raw = torch.rand(R, S, 4, generator=g)
    raw[..., 3] = torch.tanh(torch.randn(R, S, generator=g) + torch.linspace(2, -2, S))
    z = torch.sort(torch.rand(R, S, generator=g) * 4, -1)[0]
    rr = raw.clone().requires_grad_(True); br = torch.tensor([10.0], requires_grad=True)
    ref = O.composite(rr, z, br)
    probes = [torch.randn(v.shape, generator=g) for v in (ref[0], ref[1], ref[2], ref[3], ref[6])]
    sum((p * v).sum() for p, v in zip(probes, (ref[0], ref[1], ref[2], ref[3], ref[6]))).backward()
    rg = raw.to(DEV).requires_grad_(True); bg = torch.tensor([10.0], device=DEV, requires_grad=True)
    out = _CompositeFn.apply(rg, z.to(DEV), bg)[:5]
    for a, b in zip(out, (ref[0], ref[1], ref[2], ref[3], ref[6])):
        close(a, b, 2e-5, 2e-6)
    sum((p.to(DEV) * v).sum() for p, v in zip(probes, out)).backward()
    close(rg.grad, rr.grad, 3e-4, 3e-5)
    close(bg.grad, br.grad, 3e-4, 3e-3)


def test_sdf_losses_golden(us, golden):
    g = golden("g5_losses")
    sdf = T(g["sdf"]).to(DEV).requires_grad_(True)
    z, gt, tr = T(g["z"]).to(DEV), T(g["gt"]).to(DEV), float(g["truncation"])
    l = us.sdf_losses(sdf, z, gt, tr, 5, 200, 10)
    close(l, g["loss_map"], 1e-5, 1e-6)
    l.backward()
    close(sdf.grad, g["dsdf"], 1e-4, 1e-7)
    close(us.sdf_losses(sdf.detach(), z, gt, tr, 10, 200, 50), g["loss_trk"], 1e-5, 1e-6)
    ln = us.sdf_losses(sdf.detach(), T(g["z_far"]).to(DEV), gt, tr, 5, 200, 10)
    assert torch.isnan(ln) and np.isnan(g["loss_nan"])                           # empty selection -> NaN like torch.mean([])


@pytest.mark.parametrize("kind,mode", [("mapping", "original"), ("mapping", "no_mask"), ("tracking", "original"), ("tracking", "no_mask")])
def test_full_loss_vs_oracle(us, kind, mode):
    g = torch.Generator().manual_seed(13)
    R, S, tr = 200, 40, 0.06
    gt = torch.rand(R, generator=g) * 2 + 0.4
    if kind == "mapping":
        gt[::9] = 0.0
    z = O.sample_z_with_depth(gt.clamp(min=0.3)[:, None], tr, 32, 8, True, torch.rand(R, S, generator=g))
    sdf = torch.tanh(torch.randn(R, S, generator=g))
    depth = gt + torch.randn(R, generator=g) * 0.05; depth[3] += 30.0
    rgb = torch.rand(R, 3, generator=g); gc = torch.rand(R, 3, generator=g)
    unc = torch.rand(R, generator=g) * 0.01; unc[::5] = 0.5
    w = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
    leaves = [t.clone().requires_grad_(True) for t in (sdf, depth, rgb)]
    ret = (None, unc, leaves[1], leaves[2], leaves[0], z, None)
    lr = (O.mapping_loss if kind == "mapping" else O.tracking_loss)(ret, gt, gc, tr, w, mode)
    lr.backward()
    gl = [t.to(DEV).requires_grad_(True) for t in (sdf, depth, rgb)]
    l = us.fused_loss(kind, mode, gl[0], z.to(DEV), gl[1], gl[2], unc.to(DEV), gt.to(DEV), gc.to(DEV), tr, w)
    close(l, lr, 2e-5, 1e-6)
    l.backward()
    for a, b in zip(gl, leaves):
        close(a.grad, b.grad, 2e-4, 1e-7)


# ---------------------------------------------------------------------------------------------- rays
def test_gather_rays_golden(us, golden):
    g = golden("g1_rays")
    H, W, fx, fy, cx, cy = g["intr"]
    dev = lambda k: T(g[k]).to(DEV)
    sa = us.common.get_samples_all(0, int(H), 0, int(W), 6, int(H), int(W), fx, fy, cx, cy, dev("c2w"), dev("pool_d"),
                                   dev("pool_c"), DEV, dev("pool_r"), indices=dev("sa_idx"))
    for a, k in zip(sa, ("sa_o", "sa_d", "sa_depth", "sa_color")):
        close(a, g[k], 1e-6, 1e-6)
    ro, rd = us.common.get_rays(int(H), int(W), fx, fy, cx, cy, dev("c2w")[0], DEV)
    close(ro, g["rays_o"], 1e-6, 1e-6); close(rd, g["rays_d"], 1e-6, 1e-6)
    # bbox filter against the reference expression (Mapper.py:396-402)
    o, d = T(g["sa_o"]), T(g["sa_d"])
    far = O.bbox_far(o, d, BOUND)
    gt = far * torch.tensor([0.5, 1.5] * (o.shape[0] // 2))
    m = us.common.bbox_filter(o.to(DEV), d.to(DEV), gt.to(DEV), BOUND)
    assert torch.equal(m.cpu(), far >= gt)
    close(us.common.bbox_far(o.to(DEV), d.to(DEV), BOUND), far, 1e-6, 1e-6)


def test_get_samples_golden(us, golden, monkeypatch):
    """common.get_samples -> get_sample_uv -> select_uv -> get_rays_from_uv (src/common.py:168-180,133-150,109-131,95-107) on the
    GPU against the reference's own outputs (fixture g1: s1 = one frame with a crop region, s3 = three frames, whole image).  The
    only random draw of the chain, torch.randint at common.py:116, is replaced by the draw the fixture recorded."""
    g = golden("g1_rays")
    H, W, fx, fy, cx, cy = g["intr"]; H, W = int(H), int(W)
    dev = lambda k: T(g[k]).to(DEV)
    for tag, (H0, H1, W0, W1), n, b in (("s1", (2, H - 2, 3, W - 3), 7, 1), ("s3", (0, H, 0, W), 5, 3)):
        idx = dev(tag + "_idx")
        monkeypatch.setattr(torch, "randint", lambda *a, **k: idx)
        out = us.common.get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, dev("c2w")[:b], dev("depths")[:b], dev("colors")[:b], DEV)
        monkeypatch.undo()
        for a, k in zip(out, ("o", "d", "depth", "color")):
            assert a.is_cuda
            close(a, g[f"{tag}_{k}"], 1e-6, 1e-6)


def test_g4_decoders_on_the_fused_mlp(us, golden):
    """the reference's torch-MLP Decoders (src/networks/decoders.py:74-84,122-128,147-153: 32 -> 16 -> 16 -> out, biases, tanh /
    sigmoid) with its own state_dict: outputs, input gradients and every parameter gradient of fixture g4, reproduced by the
    fused kernels (us_mlp_fwd / us_mlp_bwd through Decoders.get_raw_sdf / get_raw_rgb)."""
    g = golden("g4_decoders")
    dec = us.Decoders(_cfg(), c_dim=32, truncation=0.06, learnable_beta=True)
    sd = {k.replace("__", "."): T(g[k]) for k in g if (k.startswith(("linears", "c_linears", "output_linear", "c_output_linear")) or k == "beta")}
    dec.load_state_dict(sd)
    dec = dec.to(DEV)
    fs = T(g["feat_s"]).to(DEV).requires_grad_(True); fc = T(g["feat_c"]).to(DEV).requires_grad_(True)
    sr = ([lambda p: fs], [lambda p: fc])
    p = torch.rand(fs.shape[0], 3, device=DEV)
    sdf = dec.get_raw_sdf(p, sr); rgb = dec.get_raw_rgb(p, sr)
    close(sdf, g["sdf"], 2e-5, 1e-6); close(rgb, g["rgb"], 2e-5, 1e-6)
    ((sdf * T(g["probe_s"]).to(DEV)).sum() + (rgb * T(g["probe_c"]).to(DEV)).sum()).backward()
    close(fs.grad, g["dfeat_s"], 1e-4, 1e-6); close(fc.grad, g["dfeat_c"], 1e-4, 1e-6)
    n_checked = 0
    for n, p_ in dec.named_parameters():
        k = "grad__" + n.replace(".", "__")
        if k in g:
            close(p_.grad, g[k], 1e-4, 1e-5); n_checked += 1
    assert n_checked == 12


# ---------------------------------------------------------------------------------------------- end to end
def _grid(us, params, log2T=10, res=64):
    enc = us.HashGridEncoding(3, enc_cfg(log2T, res)).to(DEV)
    with torch.no_grad():
        enc.params.copy_(T(params))
    return enc


def _cfg(n_strat=32, n_imp=8, perturb=True, tcnn=False):
    return {"rendering": {"perturb": perturb, "n_stratified": n_strat, "n_importance": n_imp}, "scale": 1,
            "grid_mode": "hash_grid", "grid": {"tcnn_network": tcnn}, "model": {"c_dim": 32, "truncation": 0.06}}


def _renderer(us, cfg, H=12, W=16, fx=10., fy=10., cx=7.5, cy=5.5):
    import types
    return us.Renderer(cfg, types.SimpleNamespace(bound=BOUND, device=DEV, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy))


def _decoders(us, g, prefix, cfg):
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
    sd = {k[len(prefix):].replace("__", "."): T(v) for k, v in g.items() if k.startswith(prefix)}
    dec.load_state_dict(sd)                       # reference state_dict keys load unchanged
    return dec.to(DEV)


def test_g7_render_batch_ray_fwd_bwd(us, golden):
    g = golden("g7_render")
    cfg = _cfg()
    dec = _decoders(us, g, "dec__", cfg)
    es, ec = _grid(us, g["grid_s"]), _grid(us, g["grid_c"])
    ro = T(g["rays_o"]).to(DEV).requires_grad_(True); rd = T(g["rays_d"]).to(DEV).requires_grad_(True)
    torch.manual_seed(int(g["seed"]))
    t_rand = torch.rand(ro.shape[0], 40)            # the reference's CPU draw for this seed
    r = _renderer(us, cfg)
    term, unc, depth, rgb, sdf, z, dunc = r.render_batch_ray(([es], [ec]), dec, rd, ro, DEV, 0.06, gt_depth=T(g["gt_depth"]).to(DEV),
                                                            t_rand=t_rand.to(DEV))
    assert np.array_equal(z.cpu().numpy(), g["z_vals"])                              # bit-exact samples
    outs = dict(term=term, unc=unc, depth=depth, rgb=rgb, sdf=sdf, dunc=dunc)
    for k, v in outs.items():
        close(v, g[k], 1e-3, 1e-5)                  # north_star tolerance: 1e-3 relative on rendered RGB / depth
        close(v, g[k], 5e-5, 5e-6)                  # what the fp32 path actually holds
    sum((T(g["probe_" + k]).to(DEV) * v).sum() for k, v in outs.items()).backward()
    close(es.params.grad, g["g_grid_s"], 2e-4, 2e-6); close(ec.params.grad, g["g_grid_c"], 2e-4, 2e-6)
    close(ro.grad, g["g_rays_o"], 2e-3, 2e-3); close(rd.grad, g["g_rays_d"], 2e-3, 2e-3)
    for n, p_ in dec.named_parameters():
        ref = g["gdec__" + n.replace(".", "__")]
        close(p_.grad, ref, 2e-4, 1e-5 * max(1.0, float(np.abs(ref).max())))


def test_g6_zero_depth_branch(us, golden):
    g = golden("g6_zerodepth")
    cfg = _cfg()
    dec = _decoders(us, g, "dec__", cfg)
    es, ec = _grid(us, g["grid_s"]), _grid(us, g["grid_c"])
    r = _renderer(us, cfg)
    # replay the reference's CPU random stream: jitter [R1,S], coarse jitter [R0,32], pdf uniforms [R0,8]
    gt = T(g["gt_depth"])
    R1, R0 = int((gt > 0).sum()), int((gt <= 0).sum())
    torch.manual_seed(int(g["seed"]))
    tr1, tr0, u0 = torch.rand(R1, 40), torch.rand(R0, 32), torch.rand(R0, 8)
    draws = [tr0.to(DEV), u0.to(DEV)]
    real_rand = torch.rand
    try:
        torch.rand = lambda *a, **k: draws.pop(0)           # the two draws inside the zero-depth branch, in order
        ret = r.render_batch_ray(([es], [ec]), dec, T(g["rays_d"]).to(DEV), T(g["rays_o"]).to(DEV), DEV, 0.06,
                                 gt_depth=gt.to(DEV), t_rand=tr1.to(DEV))
    finally:
        torch.rand = real_rand
    close(ret[5], g["z_vals"], 1e-4, 1e-5); close(ret[2], g["depth"], 1e-3, 1e-4); close(ret[3], g["rgb"], 1e-3, 1e-4)


@pytest.mark.parametrize("mode", ["original", "no_mask"])
def test_g8_tracking_iteration(us, golden, mode):
    g = golden("g8_tracking")
    H, W, fx, fy, cx, cy = g["intr"]; H, W = int(H), int(W)
    eh, ew = int(g["edge"][0]), int(g["edge"][1])
    cfg = _cfg()
    dec = _decoders(us, g, "dec__", cfg)
    for p in dec.parameters():
        p.requires_grad_(False)
    es, ec = _grid(us, g["grid_s"]), _grid(us, g["grid_c"])
    r = _renderer(us, cfg, H, W, fx, fy, cx, cy)
    pose = T(g["pose"]).to(DEV).clone().requires_grad_(True)
    # the reference's CPU random stream: randint for the pixels, then rand for the jitter
    torch.manual_seed(int(g["seed"]))
    idx = torch.randint((H - 2 * eh) * (W - 2 * ew), (int(g["n"]),))
    c2w = us.common.cam_pose_to_matrix(pose)
    i, j = torch.meshgrid(torch.linspace(ew, W - ew - 1, W - 2 * ew), torch.linspace(eh, H - eh - 1, H - 2 * eh), indexing="ij")
    i, j = i.t().reshape(-1)[idx].to(DEV)[None], j.t().reshape(-1)[idx].to(DEV)[None]
    gd_img, gc_img = T(g["gt_depth"])[:, eh:H - eh, ew:W - ew].reshape(1, -1), T(g["gt_color"])[:, eh:H - eh, ew:W - ew].reshape(1, -1, 3)
    gd, gc = gd_img[0, idx].to(DEV), gc_img[0, idx].to(DEV)
    ro, rd = us.common.get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy, DEV)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    inside = us.common.bbox_filter(ro, rd, gd, BOUND, require_depth=True)          # Tracker.py:177-184
    ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
    t_rand = torch.rand(ro.shape[0], 40).to(DEV)
    ret = r.render_batch_ray(([es], [ec]), dec, rd, ro, DEV, 0.06, gt_depth=gd, t_rand=t_rand)
    loss = us.tracking_loss(ret, gd, gc, 0.06, dict(fs=10, center=200, tail=50, color=5, depth=1), mode)
    loss.backward()
    close(loss, g[f"{mode}_loss"], 1e-4, 1e-5); close(ret[1], g[f"{mode}_unc"], 1e-4, 1e-6)
    close(pose.grad, g[f"{mode}_gpose"], 5e-3, 5e-3)
    close(es.params.grad, g[f"{mode}_ggrid_s"], 5e-4, 1e-5)


def test_g9_mapping_two_iterations(us, golden):
    g = golden("g9_mapping")
    H, W, fx, fy, cx, cy = g["intr"]; H, W = int(H), int(W)
    cfg = _cfg()
    dec = _decoders(us, g, "dec0__", cfg)
    es, ec = _grid(us, g["grid_s0"]), _grid(us, g["grid_c0"])
    r = _renderer(us, cfg, H, W, fx, fy, cx, cy)
    gt_depth, gt_color, c2w = T(g["gt_depth"]), T(g["gt_color"]), T(g["c2w"])
    cam = O.get_camera_rays(H, W, fx, fy, cx, cy)
    torch.manual_seed(int(g["seed"]))
    idx = torch.randperm(H * W)[:int(H * W * 0.1)]
    pool_c, pool_d, pool_r = (gt_color.reshape(-1, 3)[idx][None].to(DEV), gt_depth.reshape(-1)[idx][None].to(DEV),
                              cam.reshape(-1, 3)[idx][None].to(DEV))
    f = float(g["lr_factor"])
    opt = torch.optim.Adam([{"params": list(dec.parameters()), "lr": 0.001 * f},
                            {"params": [es.params], "lr": 0.05 * f}, {"params": [ec.params], "lr": 0.05 * f}])
    w = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
    P = pool_d.shape[1]
    for _ in range(int(g["iters"])):
        indices = torch.randint(P, (int(g["pixels"]),)).reshape(1, -1)            # CPU stream of the reference
        ro, rd, gd, gc = us.common.get_samples_all(0, H, 0, W, int(g["pixels"]), H, W, fx, fy, cx, cy, c2w[None].to(DEV),
                                                   pool_d, pool_c, DEV, pool_r, indices=indices.to(DEV))
        inside = us.common.bbox_filter(ro, rd, gd, BOUND)
        ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
        n_depth = int((gd > 0).sum())
        t_rand = torch.rand(n_depth, 40)
        n_zero = ro.shape[0] - n_depth
        draws = [torch.rand(n_zero, 32).to(DEV), torch.rand(n_zero, 8).to(DEV)] if n_zero else []
        real_rand = torch.rand
        try:
            if draws:
                torch.rand = lambda *a, **k: draws.pop(0)
            ret = r.render_batch_ray(([es], [ec]), dec, rd, ro, DEV, 0.06, gt_depth=gd, t_rand=t_rand.to(DEV))
        finally:
            torch.rand = real_rand
        loss = us.mapping_loss(ret, gd, gc, 0.06, w, "original")
        opt.zero_grad(); loss.backward(); opt.step()
    close(es.params, g["grid_s1"], 1e-3, 2e-5); close(ec.params, g["grid_c1"], 1e-3, 2e-5)
    for n, v in dec.state_dict().items():
        close(v, g["dec1__" + n.replace(".", "__")], 1e-3, 2e-5)


def test_tcnn_layout_decoders_vs_oracle(us):
    """tcnn_network=True (Replica configs): FusedMLP flat params, no bias, 32->16->out, vs the fp32 oracle."""
    torch.manual_seed(5)
    cfg = _cfg(tcnn=True)
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06).to(DEV)
    assert set(dec.state_dict().keys()) == {"beta", "sdf_decoder.params", "color_decoder.params"}
    assert dec.sdf_decoder.params.numel() == 768 and dec.color_decoder.params.numel() == 768
    od = O.DecodersOracle(tcnn_network=True)
    with torch.no_grad():
        od.sdf_params.copy_(dec.sdf_decoder.params.cpu()); od.color_params.copy_(dec.color_decoder.params.cpu())
    rng = np.random.default_rng(0)
    pg = rng.standard_normal(O.make_grid_desc(16, 2, 10, 16, O.per_level_scale(64)).n_params).astype(np.float32) * 0.5
    es, ec = _grid(us, pg), _grid(us, pg[::-1].copy())
    os_, oc_ = O.HashGridOracle(3, enc_cfg(10, 64)), O.HashGridOracle(3, enc_cfg(10, 64))
    with torch.no_grad():
        os_.params.copy_(T(pg)); oc_.params.copy_(T(pg[::-1].copy()))
    p = torch.rand(7, 9, 3)
    raw = dec(p.to(DEV), ([es], [ec]))
    ref = od(p, ([os_], [oc_]))
    close(raw, ref, 1e-4, 1e-5)


def test_render_img_and_eval_points_vs_oracle(us):
    """forward-only consumers: Renderer.render_img (Renderer.py:160-223, chunked) and Mesher.eval_points (Mesher.py:134-166)"""
    import types
    torch.manual_seed(2)
    cfg = _cfg(perturb=False)
    H, Wd, fx, fy, cx, cy = 9, 13, 9.0, 9.0, 6.0, 4.0
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06).to(DEV)
    od = O.DecodersOracle(); od.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    rng = np.random.default_rng(1)
    pg = rng.standard_normal(O.make_grid_desc(16, 2, 10, 16, O.per_level_scale(64)).n_params).astype(np.float32) * 0.4
    es, ec = _grid(us, pg), _grid(us, pg[::-1].copy())
    os_, oc_ = O.HashGridOracle(3, enc_cfg(10, 64)), O.HashGridOracle(3, enc_cfg(10, 64))
    with torch.no_grad():
        os_.params.copy_(T(pg)); oc_.params.copy_(T(pg[::-1].copy()))
    c2w = O.cam_pose_to_matrix(torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]]))[0]
    gt = torch.rand(H, Wd) * 2 + 0.4
    r = us.Renderer(cfg, types.SimpleNamespace(bound=BOUND, device=DEV, H=H, W=Wd, fx=fx, fy=fy, cx=cx, cy=cy), ray_batch_size=50)
    depth, color, term, unc, dunc = r.render_img(([es], [ec]), dec, c2w.to(DEV), 0.06, DEV, gt_depth=gt.to(DEV))
    ro, rd = O.get_rays(H, Wd, fx, fy, cx, cy, c2w)
    ref = O.render_batch_ray(([os_], [oc_]), od, rd.reshape(-1, 3), ro.reshape(-1, 3), 0.06, gt.reshape(-1), BOUND, 32, 8, False)
    assert depth.dtype == torch.float64 and depth.shape == (H, Wd) and color.shape == (H, Wd, 3)
    close(depth.float(), ref[2].detach().reshape(H, Wd), 1e-3, 1e-5); close(color, ref[3].detach().reshape(H, Wd, 3), 1e-3, 1e-5)
    close(term.float(), ref[0].detach().reshape(H, Wd), 1e-3, 1e-5); close(dunc.float(), ref[6].detach().reshape(H, Wd), 2e-3, 1e-4)
    # dense query: inside points decode, outside points get sdf -1
    p = torch.rand(777, 3) * (BOUND[:, 1] - BOUND[:, 0]) * 1.2 + BOUND[:, 0] - 0.1 * (BOUND[:, 1] - BOUND[:, 0])
    out = us.eval_points(p.to(DEV), ([es], [ec]), dec, BOUND, points_batch_size=200)
    inside = ((p < BOUND[:, 1]) & (p > BOUND[:, 0])).all(-1)
    refq = od((p - BOUND[:, 0]) / (BOUND[:, 1] - BOUND[:, 0]), ([os_], [oc_])).detach()
    refq[~inside, 3] = -1
    close(out, refq, 1e-4, 1e-5)


# ---------------------------------------------------------------------------------------------- optimiser
def test_adam_matches_torch(us):
    import ctypes
    from unislam_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    n = 100003
    p0 = torch.randn(n, generator=g)
    pt = p0.clone().to(DEV).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=0.05)
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g).to(DEV) * (0.1 if step != 3 else 0.0)
        pt.grad = gr.clone(); opt.step()
        L.check(L.lib().us_adam_step(L.ptr(p), L.ptr(gr), L.ptr(m), L.ptr(v), n, 0.05, 0.9, 0.999, 1e-8, step, L.stream()), "adam")
    close(p, pt.detach(), 2e-5, 2e-6)


def test_sample_points_equals_the_three_kernels(us):
    """us_sample_points (filter + z + points in one launch) against us_bbox_filter / us_sample_z / us_ray_points, bit for bit;
    the in-kernel jitter generator yields iid-looking U[0,1) draws (mean, variance, range) and depends on the seed."""
    import ctypes
    from unislam_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    R, ns, ni = 1003, 48, 16
    S = ns + ni
    bound = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])
    bh = us.common.bound_host(bound)
    o = (torch.tensor([[3.0, 1.2, 0.0]]).repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05).to(DEV)
    d = torch.randn(R, 3, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).to(DEV)
    gd = (torch.rand(R, generator=g) * 6 + 0.2).to(DEV)                      # some beyond the box -> invalid
    tr = torch.rand(R, S, generator=g).to(DEV)
    tu, ts = torch.linspace(0., 1., ns).to(DEV), torch.linspace(0., 1., ni).to(DEV)
    cf, so, sp = ctypes.c_float(1.2), ctypes.c_float(1.5 * 0.06), ctypes.c_float(3 * 0.06)
    lib, st, P = L.lib(), L.stream(), L.ptr
    v0 = torch.empty(R, dtype=torch.uint8, device=DEV); z0 = torch.empty(R, S, device=DEV); p0 = torch.empty(R, S, 3, device=DEV)
    L.check(lib.us_bbox_filter(P(o), P(d), P(gd), bh, R, 0, P(v0), None, st), "f")
    L.check(lib.us_sample_z(P(gd), R, P(tu), ns, P(ts), ni, cf, so, sp, P(tr), P(z0), st), "z")
    L.check(lib.us_ray_points(P(o), P(d), P(z0), bh, R, S, P(p0), st), "p")
    v1 = torch.empty_like(v0); z1 = torch.empty_like(z0); p1 = torch.empty_like(p0)
    L.check(lib.us_sample_points(P(o), P(d), P(gd), bh, R, P(tu), ns, P(ts), ni, cf, so, sp, P(tr), 0, None, 1, 0, P(v1), P(z1), P(p1), st), "sp")
    assert torch.equal(v0, v1) and torch.equal(z0, z1) and torch.equal(p0, p1)
    assert 0 < int(v1.sum()) < R
    # unperturbed
    L.check(lib.us_sample_z(P(gd), R, P(tu), ns, P(ts), ni, cf, so, sp, None, P(z0), st), "z")
    L.check(lib.us_sample_points(P(o), P(d), P(gd), bh, R, P(tu), ns, P(ts), ni, cf, so, sp, None, 0, None, 0, 0, P(v1), P(z1), P(p1), st), "sp")
    assert torch.equal(z0, z1)
    # in-kernel generator: recover u from z = lower + (upper - lower) * u
    mids = 0.5 * (z0[:, 1:] + z0[:, :-1])
    lower = torch.cat([z0[:, :1], mids], -1); upper = torch.cat([mids, z0[:, -1:]], -1)
    us_ = []
    for seed in (1, 2):
        L.check(lib.us_sample_points(P(o), P(d), P(gd), bh, R, P(tu), ns, P(ts), ni, cf, so, sp, None, seed, None, 1, 0, P(v1), P(z1), P(p1), st), "sp")
        ok = (upper - lower) > 1e-4
        u = ((z1 - lower) / (upper - lower))[ok]
        assert float(u.min()) >= -1e-3 and float(u.max()) <= 1 + 1e-3
        assert abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1 / 12) < 3e-3
        us_.append(z1.clone())
    assert not torch.equal(us_[0], us_[1])


def test_adam_segments_equals_adam_per_segment(us):
    import ctypes
    from unislam_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    n = 100000
    p0 = torch.randn(n, generator=g).to(DEV); gr = torch.randn(n, generator=g).to(DEV)
    m0 = torch.randn(n, generator=g).to(DEV) * 0.1; v0 = torch.rand(n, generator=g).to(DEV) * 0.01
    segs = [(0, 1000, 1e-3), (1024, 40000, 5e-2), (50000, 50000, 2e-2)]
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    lib, st, P = L.lib(), L.stream(), L.ptr
    off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
    for (o, k, lr) in segs:
        L.check(lib.us_adam_step(off(pa, o), off(gr, o), off(ma, o), off(va, o), k, lr, 0.9, 0.999, 1e-8, 3, st), "adam")
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    I64, DBL = ctypes.c_int64 * 3, ctypes.c_double * 3
    L.check(lib.us_adam_step_segments(P(pb), P(gr), P(mb), P(vb), 3, I64(*[s[0] for s in segs]), I64(*[s[1] for s in segs]),
                                      DBL(*[s[2] for s in segs]), 0.9, 0.999, 1e-8, 3, 0b101, st), "adam segs")
    assert float(gr[:1000].abs().max()) == 0.0 and float(gr[50000:].abs().max()) == 0.0 and float(gr[1024:41024].abs().min()) > 0.0
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.equal(pb[1000:1024], p0[1000:1024])                      # gaps between segments are left alone


@pytest.mark.parametrize("log2T", [14, 19])
def test_input_gradient_by_regathering_equals_stored_dydx(us, log2T):
    """us_hashgrid_bwd_input_gather (no stored dy_dx) against us_hashgrid_fwd(dy_dx) + us_hashgrid_bwd_input (to rounding);
    both dL_dy layouts, the clamp flag, accumulation; and against the oracle."""
    import ctypes
    from unislam_amd import _lib as L
    g = torch.Generator(device=DEV).manual_seed(log2T)
    n = 5003
    x = torch.rand((n, 3), device=DEV, generator=g) * 1.2 - 0.1                 # some coordinates outside [0, 1]
    enc = us.HashGridEncoding(3, enc_cfg(log2T)).to(DEV)
    with torch.no_grad():
        enc.params.copy_(torch.randn(enc.params.shape, device=DEV, generator=g) * 0.2)
    p = enc.params.detach()
    dy = torch.randn((n, 32), device=DEV, generator=g)
    dy_lm = dy.view(n, 16, 2).permute(1, 0, 2).contiguous()
    lib, st, P = L.lib(), L.stream(), L.ptr
    d = ctypes.byref(enc.desc)
    for clamp in (0, 1):
        out = torch.empty((n, 32), device=DEV); dydx = torch.empty((n, 32, 3), device=DEV)
        L.check(lib.us_hashgrid_fwd(d, P(p), P(x), n, P(out), P(dydx), clamp, st), "fwd")
        ref = torch.empty((n, 3), device=DEV)
        L.check(lib.us_hashgrid_bwd_input(P(dy), P(dydx), n, 32, P(ref), st), "bwd_input")
        a = torch.empty((n, 3), device=DEV); b = torch.empty((n, 3), device=DEV)
        L.check(lib.us_hashgrid_bwd_input_gather(d, P(p), P(x), P(dy), n, P(a), clamp, st), "gather")
        L.check(lib.us_hashgrid_bwd_input_gather(d, P(p), P(x), P(dy_lm), n, P(b), clamp | L.US_GRID_LEVEL_MAJOR, st), "gather lm")
        # level-parallel sum: the 16 per-level contributions are added in level order, a different association than the
        # channel-by-channel sum of the stored path -> equal to rounding, and the two layouts bit-identical
        assert torch.allclose(a, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max())) and torch.equal(a, b)
        L.check(lib.us_hashgrid_bwd_input_gather(d, P(p), P(x), P(dy), n, P(b), clamp | L.US_GRID_ACCUMULATE, st), "gather acc")
        assert torch.equal(b, 2 * a)                                        # accumulate: b held a, a was added once more
        if clamp:
            outside = ((x < 0) | (x > 1))
            assert float(a[outside].abs().max()) == 0.0
    xin = x.clamp(0, 1).cpu().numpy()
    dsc = O.make_grid_desc(16, 2, log2T, 16, PLS816)
    _, dydx_o = O.hashgrid_fwd(dsc, p.cpu().numpy(), xin, True)
    gx = O.hashgrid_bwd_input(dy.cpu().numpy(), dydx_o) * ((x >= 0) & (x <= 1)).cpu().numpy()
    np.testing.assert_allclose(a.cpu().numpy(), gx, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("log2T,n", [(16, 70001), (19, 20000)])
def test_forward_counted_feeds_the_binned_backward(us, log2T, n):
    """us_hashgrid_fwd_counted = us_hashgrid_fwd bit for bit + the counts of the binning; us_hashgrid_bwd_binned(COUNTED) on them
    gives the gradient of the counting backward (dead samples -- zero gradient rows -- become zero records)."""
    import ctypes
    from unislam_amd import _lib as L
    g = torch.Generator(device=DEV).manual_seed(n)
    # ray-like order (runs of samples in one cell) + some out-of-range coordinates for the clamp
    R = n // 50 + 1
    o = torch.rand((R, 1, 3), device=DEV, generator=g) * 0.8 + 0.1
    dirs = torch.randn((R, 1, 3), device=DEV, generator=g) * 0.2
    t = torch.linspace(0, 1, 50, device=DEV).reshape(1, 50, 1)
    x = (o + dirs * t).reshape(-1, 3)[:n].contiguous()
    enc = us.HashGridEncoding(3, enc_cfg(log2T)).to(DEV)
    with torch.no_grad():
        enc.params.copy_(torch.randn(enc.params.shape, device=DEV, generator=g) * 0.2)
    p = enc.params.detach()
    lib, st, P = L.lib(), L.stream(), L.ptr
    d = ctypes.byref(enc.desc)
    nbytes = int(lib.us_hashgrid_bwd_workspace_bytes(d, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    for flags in (1, 3):
        shape = (n, 32) if flags == 1 else (16, n, 2)
        ref = torch.empty(shape, device=DEV); out = torch.empty(shape, device=DEV)
        L.check(lib.us_hashgrid_fwd(d, P(p), P(x), n, P(ref), None, flags, st), "fwd")
        L.check(lib.us_hashgrid_fwd_counted(d, P(p), P(x), n, P(out), flags, P(ws), nbytes, st), "fwd counted")
        assert torch.equal(out, ref)
        dy = torch.randn(shape, device=DEV, generator=g)
        if flags == 1:
            dy[::9] = 0.0
        else:
            dy[:, ::9] = 0.0
        g0 = torch.empty(enc.desc.n_params, device=DEV); g1 = torch.full((enc.desc.n_params,), 7.0, device=DEV)
        ws0 = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        L.check(lib.us_hashgrid_bwd_binned(d, P(x), P(dy), n, P(g0), flags | L.US_GRID_BWD_OVERWRITE, P(ws0), nbytes, st), "bwd")
        L.check(lib.us_hashgrid_bwd_binned(d, P(x), P(dy), n, P(g1), flags | L.US_GRID_BWD_OVERWRITE | L.US_GRID_BWD_COUNTED, P(ws), nbytes, st),
                "bwd counted")
        assert torch.allclose(g1, g0, rtol=1e-6, atol=1e-7 * float(g0.abs().max()))
        # the scan passes ahead of the gradient call (us_hashgrid_bwd_scan, then COUNTED | SCANNED): the same gradient (split bins are
        # summed with float atomics, so up to their order)
        L.check(lib.us_hashgrid_fwd_counted(d, P(p), P(x), n, P(out), flags, P(ws), nbytes, st), "fwd counted")
        g2 = torch.full((enc.desc.n_params,), -3.0, device=DEV)
        L.check(lib.us_hashgrid_bwd_scan(d, n, P(g2), flags | L.US_GRID_BWD_OVERWRITE, P(ws), nbytes, st), "scan")
        L.check(lib.us_hashgrid_bwd_binned(d, P(x), P(dy), n, P(g2), flags | L.US_GRID_BWD_OVERWRITE | L.US_GRID_BWD_COUNTED | L.US_GRID_BWD_SCANNED,
                                           P(ws), nbytes, st), "bwd scanned")
        assert torch.allclose(g2, g1, rtol=1e-6, atol=1e-7 * float(g1.abs().max()))
        # SCANNED without COUNTED is a configuration error
        assert lib.us_hashgrid_bwd_binned(d, P(x), P(dy), n, P(g2), flags | L.US_GRID_BWD_SCANNED, P(ws), nbytes, st) == L.US_ERR_CONFIG


@pytest.mark.parametrize("F,n", [(1, 333), (1, 40000), (4, 333), (4, 40000)])
def test_binned_backward_other_feature_widths(us, F, n):
    """the binned table gradient for F = 1 (staged records with one value) and F = 4 (records stored straight from the lanes),
    ray-ordered points so that runs get combined; also the counting forward pass for these widths"""
    import ctypes
    from unislam_amd import _lib as L
    rng = np.random.default_rng(F * 100 + n % 97)
    R = n // 40 + 1
    o = rng.random((R, 1, 3), dtype=np.float32) * 0.8 + 0.1
    dirs = (rng.standard_normal((R, 1, 3)) * 0.15).astype(np.float32)
    x = np.clip((o + dirs * np.linspace(0, 1, 40, dtype=np.float32).reshape(1, 40, 1)).reshape(-1, 3)[:n], 0, 1).astype(np.float32)
    e = us.HashGridEncoding(3, enc_cfg(12, res=300, L=8, F=F)).to(DEV)
    d = O.make_grid_desc(8, F, 12, 16, O.per_level_scale(300))
    p = rng.standard_normal(d.n_params).astype(np.float32)
    with torch.no_grad():
        e.params.copy_(T(p))
    dy = rng.standard_normal((n, 8 * F)).astype(np.float32)
    dy[::11] = 0.0
    gp = O.hashgrid_bwd_params(d, x, dy)
    xd, dyd, pd = T(x).to(DEV), T(dy).to(DEV), e.params.detach()
    lib, st, P = L.lib(), L.stream(), L.ptr
    dd = ctypes.byref(e.desc)
    nbytes = int(lib.us_hashgrid_bwd_workspace_bytes(dd, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    g = torch.full((d.n_params,), 3.0, device=DEV)
    L.check(lib.us_hashgrid_bwd_binned(dd, P(xd), P(dyd), n, P(g), L.US_GRID_BWD_OVERWRITE, P(ws), nbytes, st), "binned")
    np.testing.assert_allclose(g.cpu().numpy(), gp, rtol=1e-4, atol=2e-6 * np.abs(gp).max())
    out = torch.empty((n, 8 * F), device=DEV); ref = torch.empty((n, 8 * F), device=DEV)
    L.check(lib.us_hashgrid_fwd(dd, P(pd), P(xd), n, P(ref), None, 0, st), "fwd")
    L.check(lib.us_hashgrid_fwd_counted(dd, P(pd), P(xd), n, P(out), 0, P(ws), nbytes, st), "fwd counted")
    assert torch.equal(out, ref)
    g2 = torch.full((d.n_params,), -1.0, device=DEV)
    L.check(lib.us_hashgrid_bwd_binned(dd, P(xd), P(dyd), n, P(g2), L.US_GRID_BWD_OVERWRITE | L.US_GRID_BWD_COUNTED, P(ws), nbytes, st), "counted")
    np.testing.assert_allclose(g2.cpu().numpy(), gp, rtol=1e-4, atol=2e-6 * np.abs(gp).max())


def test_backward_of_a_table_beyond_the_bin_budget_falls_back(us):
    """log2T = 22: more than 4096 bins of 2048 entries -> us_hashgrid_bwd_binned answers US_ERR_CONFIG and the module's automatic
    mode takes the sliced kernels; the gradient still satisfies the adjoint identity"""
    g = torch.Generator(device=DEV).manual_seed(5)
    n = 20000
    enc = us.HashGridEncoding(3, enc_cfg(22)).to(DEV)
    with torch.no_grad():
        enc.params.copy_(torch.randn(enc.params.shape, device=DEV, generator=g) * 0.1)
    x = torch.rand((n, 3), device=DEV, generator=g)
    dy = torch.randn((n, 32), device=DEV, generator=g)
    out = enc(x)
    out.backward(dy)
    lhs = (dy.double() * out.detach().double()).sum()
    rhs = (enc.params.grad.double() * enc.params.detach().double()).sum()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-2


@pytest.mark.parametrize("n", [1, 2, 7, 2000, 5000, 8192])
def test_masked_median_equals_torch_median(us, n):
    """us_masked_median = torch.median(|a - b|[valid]) (lower median), the tracking loss's 10 x median gate (Tracker.py:212-214)"""
    from unislam_amd import _lib as L
    g = torch.Generator().manual_seed(n)
    a, b = torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    valid = (torch.rand(n, generator=g) > 0.3).to(torch.uint8).to(DEV)
    valid[0] = 1
    out = torch.empty(1, device=DEV)
    L.check(L.lib().us_masked_median(L.ptr(a), L.ptr(b), L.ptr(valid), n, L.ptr(out), L.stream()), "median")
    ref = (a - b).abs()[valid.bool()].median()
    assert float(out) == float(ref)
    L.check(L.lib().us_masked_median(L.ptr(a), L.ptr(b), None, n, L.ptr(out), L.stream()), "median")
    assert float(out) == float((a - b).abs().median())
    valid.zero_()
    L.check(L.lib().us_masked_median(L.ptr(a), L.ptr(b), L.ptr(valid), n, L.ptr(out), L.stream()), "median")
    assert float(out) == float("inf")


@pytest.mark.parametrize("Su,ni", [(32, 8), (48, 8), (80, 16), (128, 64), (3, 1)])
def test_importance_z_equals_the_reference_chain(us, Su, ni):
    """us_importance_z (zero-depth rays: alpha -> weights -> un-normalised cdf -> inverse transform -> sort, one launch) against the
    oracle's restatement of Renderer.py:121-130 + common.sample_pdf on the same draws"""
    from unislam_amd import _lib as L
    g = torch.Generator().manual_seed(Su * 100 + ni)
    R = 777
    z = torch.sort(torch.rand(R, Su, generator=g) * 4 + 0.05, -1)[0]
    sdf = (torch.rand(R, Su, generator=g) * 2 - 1) * 0.5
    sdf[::5] = torch.linspace(0.6, -0.6, Su)                         # a clean surface crossing
    sdf[1::9] = 0.9                                                  # free space only: cdf ~ 0 -> the denom < 1e-5 branch
    u = torch.rand(R, ni, generator=g)
    beta = torch.tensor([7.3])
    alpha = O.sdf2alpha(sdf, beta)
    w = O.alpha_to_weights(alpha)
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    samples = O.sample_pdf(mids, w[..., 1:-1], ni, u=u) if Su > 2 else None
    ref = torch.sort(torch.cat([z, samples], -1), -1)[0]
    out = torch.empty(R, Su + ni, device=DEV)
    sdf_d, z_d, beta_d, u_d = sdf.to(DEV), z.to(DEV), beta.to(DEV), u.to(DEV)        # keep the device copies alive across the launch
    L.check(L.lib().us_importance_z(L.ptr(sdf_d), L.ptr(z_d), L.ptr(beta_d), L.ptr(u_d), R, Su, ni, L.ptr(out), L.stream()), "us_importance_z")
    o = out.cpu()
    assert bool((o[:, 1:] >= o[:, :-1]).all())
    assert torch.allclose(o, ref, rtol=2e-5, atol=2e-5), float((o - ref).abs().max())


def test_zero_depth_rows_uniform_points_and_scatter(us):
    """the pieces of the sync-light zero-depth branch: us_zero_depth_rows (ordered compaction, several 1024-row chunks),
    us_uniform_points against the torch chain of Renderer.py:106-114 on the same draws, us_importance_z_rows writing rows + points"""
    from unislam_amd import _lib as L
    from unislam_amd.common import bound_host
    lib, st, P = L.lib(), L.stream(), L.ptr
    g = torch.Generator().manual_seed(5)
    R, Su, ni = 2600, 32, 8
    bound = torch.tensor([[-2.0, 2.5], [-1.5, 2.0], [-1.0, 3.0]])
    bh = bound_host(bound)
    o = (torch.rand(R, 3, generator=g) - 0.5) * 0.8
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    gd = torch.rand(R, generator=g) * 2
    gd[torch.rand(R, generator=g) < 0.3] = 0.0
    gd[7] = float("nan")                                           # ~(gt > 0), as the reference's mask
    o_d, d_d, gd_d = o.to(DEV), d.to(DEV), gd.to(DEV)
    rows, cnt = torch.full((R,), -1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    L.check(lib.us_zero_depth_rows(P(gd_d), R, P(rows), P(cnt), st), "rows")
    want = torch.nonzero(~(gd > 0)).flatten()
    n0 = int(cnt.item())
    assert n0 == want.numel() and torch.equal(rows[:n0].cpu().long(), want)
    # coarse pass
    t_uni = torch.linspace(0., 1., Su)
    tr = torch.rand(n0, Su, generator=g)
    far = O.bbox_far(o[want], d[want], bound).unsqueeze(-1) + 0.01
    z = 0.0 * (1. - t_uni) + far * t_uni
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
    z = lower + (upper - lower) * tr
    pts = o[want].unsqueeze(1) + d[want].unsqueeze(1) * z.unsqueeze(-1)
    pts = ((pts - bound[:, 0]) / (bound[:, 1] - bound[:, 0])) * 2 - 1.0
    t_uni_d, tr_d = t_uni.to(DEV), tr.to(DEV)
    z_d, pts_d = torch.empty(n0, Su, device=DEV), torch.empty(n0, Su, 3, device=DEV)
    L.check(lib.us_uniform_points(P(o_d), P(d_d), P(rows), n0, bh, P(t_uni_d), Su, P(tr_d), 0, 1, P(z_d), P(pts_d), st), "uniform")
    assert torch.allclose(z_d.cpu(), z, rtol=1e-6, atol=1e-6) and torch.allclose(pts_d.cpu(), pts, rtol=1e-5, atol=1e-5)
    # in-kernel draws: inside the strata, different from call to call with the seed
    za, zb = torch.empty_like(z_d), torch.empty_like(z_d)
    L.check(lib.us_uniform_points(P(o_d), P(d_d), P(rows), n0, bh, P(t_uni_d), Su, None, 11, 1, P(za), P(pts_d), st), "uniform")
    L.check(lib.us_uniform_points(P(o_d), P(d_d), P(rows), n0, bh, P(t_uni_d), Su, None, 12, 1, P(zb), P(pts_d), st), "uniform")
    assert bool((za >= lower.to(DEV) - 1e-6).all()) and bool((za <= upper.to(DEV) + 1e-6).all()) and not torch.equal(za, zb)
    frac = ((za - lower.to(DEV)) / (upper - lower).to(DEV))[:, 1:-1]
    assert abs(float(frac.mean()) - 0.5) < 0.01
    # scatter: rows of the full matrix rewritten (z and unit-cube points), the others untouched
    sdf = (torch.rand(n0, Su, generator=g) - 0.5)
    u = torch.rand(n0, ni, generator=g)
    beta = torch.tensor([6.0])
    sdf_d, u_d, beta_d = sdf.to(DEV), u.to(DEV), beta.to(DEV)
    S = Su + ni
    dense = torch.empty(n0, S, device=DEV)
    L.check(lib.us_importance_z(P(sdf_d), P(z_d), P(beta_d), P(u_d), n0, Su, ni, P(dense), st), "imp")
    zf, pf = torch.full((R, S), -7.0, device=DEV), torch.full((R, S, 3), -7.0, device=DEV)
    L.check(lib.us_importance_z_rows(P(sdf_d), P(z_d), P(beta_d), P(u_d), 0, n0, Su, ni, P(rows), P(zf), P(o_d), P(d_d), bh, P(pf), st), "imp rows")
    assert torch.equal(zf[want.to(DEV)], dense)
    keep = torch.ones(R, dtype=torch.bool); keep[want] = False
    assert bool((zf[keep.to(DEV)] == -7.0).all()) and bool((pf[keep.to(DEV)] == -7.0).all())
    ref_pts = torch.empty(n0, S, 3, device=DEV)
    oc, dc = o_d[want.to(DEV)].contiguous(), d_d[want.to(DEV)].contiguous()
    L.check(lib.us_ray_points(P(oc), P(dc), P(dense), bh, n0, S, P(ref_pts), st), "pts")
    assert torch.equal(pf[want.to(DEV)], ref_pts)


@pytest.mark.parametrize("width,n_hidden,bias,prec,n", [(32, 2, True, "bf16", 70001), (32, 2, False, "bf16_plain", 4096), (16, 1, True, "bf16", 100),
                                                        (64, 2, True, "bf16", 5000), (64, 1, False, "bf16_plain", 1)])
def test_mlp_pair_equals_two_launches(us, width, n_hidden, bias, prec, n):
    """us_mlp_fwd_pair / us_mlp_bwd_pair (two decoders of one shape in one launch, blockIdx.y = decoder) against us_mlp_fwd / us_mlp_bwd
    per decoder: outputs, input gradients and parameter gradients bit-identical, with the level-major feature planes and the raw[N][4]
    output layout of the render path."""
    import ctypes
    from unislam_amd import _lib as L
    lib, st, P = L.lib(), L.stream(), L.ptr
    g = torch.Generator(device=DEV).manual_seed(width + n)
    ma = us.make_mlp_desc(32, width, n_hidden, 1, "none", bias, prec)
    mb = us.make_mlp_desc(32, width, n_hidden, 3, "sigmoid", bias, prec)
    A, B = ctypes.byref(ma), ctypes.byref(mb)
    assert lib.us_mlp_pair_supported(A, B) == 1
    pa = (torch.rand(us.network.mlp_n_params(ma), device=DEV, generator=g) * 2 - 1) * 0.4
    pb = (torch.rand(us.network.mlp_n_params(mb), device=DEV, generator=g) * 2 - 1) * 0.4
    fa, fb = torch.randn((16, n, 2), device=DEV, generator=g), torch.randn((16, n, 2), device=DEV, generator=g)
    d_raw = torch.randn((n, 4), device=DEV, generator=g)
    off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
    wsb = max(int(lib.us_mlp_bwd_workspace_bytes(A)), int(lib.us_mlp_bwd_workspace_bytes(B)))
    res = {}
    for pair in (False, True):
        raw = torch.full((n, 4), -3.0, device=DEV)
        dfa, dfb = torch.zeros_like(fa), torch.zeros_like(fb)
        ga, gb = torch.zeros_like(pa), torch.zeros_like(pb)
        wa, wb = torch.empty(wsb, dtype=torch.uint8, device=DEV), torch.empty(wsb, dtype=torch.uint8, device=DEV)
        if pair:
            L.check(lib.us_mlp_fwd_pair(A, B, P(pa), P(pb), P(fa), P(fb), n, off(raw, 3), 4, P(raw), 4, 1, st), "fwd pair")
            L.check(lib.us_mlp_bwd_pair(A, B, P(pa), P(pb), P(fa), P(fb), off(raw, 3), 4, P(raw), 4, off(d_raw, 3), 4, P(d_raw), 4, n, P(dfa), P(dfb),
                                        P(ga), P(gb), 1, P(wa), P(wb), wsb, st), "bwd pair")
        else:
            L.check(lib.us_mlp_fwd(A, P(pa), P(fa), n, off(raw, 3), 4, 1, st), "fwd a")
            L.check(lib.us_mlp_fwd(B, P(pb), P(fb), n, P(raw), 4, 1, st), "fwd b")
            L.check(lib.us_mlp_bwd(A, P(pa), P(fa), off(raw, 3), 4, off(d_raw, 3), 4, n, P(dfa), P(ga), 1, P(wa), wsb, st), "bwd a")
            L.check(lib.us_mlp_bwd(B, P(pb), P(fb), P(raw), 4, P(d_raw), 4, n, P(dfb), P(gb), 1, P(wb), wsb, st), "bwd b")
        res[pair] = (raw, dfa, dfb, ga, gb)
    for x, y in zip(res[True][:3], res[False][:3]):               # outputs, input gradients: bit-identical
        assert torch.equal(x, y)
    for x, y in zip(res[True][3:], res[False][3:]):               # parameter gradients: the same per-workgroup partials, fewer rows to sum
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6 * float(y.abs().max()))
    # the deferred reduction of the pair
    ga2, gb2 = torch.zeros_like(pa), torch.zeros_like(pb)
    wa, wb = torch.empty(wsb, dtype=torch.uint8, device=DEV), torch.empty(wsb, dtype=torch.uint8, device=DEV)
    L.check(lib.us_mlp_bwd_pair(A, B, P(pa), P(pb), P(fa), P(fb), off(res[True][0], 3), 4, P(res[True][0]), 4, off(d_raw, 3), 4, P(d_raw), 4, n,
                                None, None, P(ga2), P(gb2), 1 | 2, P(wa), P(wb), wsb, st), "bwd pair, deferred")
    assert float(ga2.abs().max()) == 0.0 and float(gb2.abs().max()) == 0.0
    L.check(lib.us_mlp_reduce_pair(A, B, P(wa), P(wb), wsb, n, P(ga2), P(gb2), st), "reduce pair")
    assert torch.equal(ga2, res[True][3]) and torch.equal(gb2, res[True][4])
    # input gradients only (tracking): no parameter gradients, no workspaces
    dfa, dfb = torch.zeros_like(fa), torch.zeros_like(fb)
    L.check(lib.us_mlp_bwd_pair(A, B, P(pa), P(pb), P(fa), P(fb), off(res[True][0], 3), 4, P(res[True][0]), 4, off(d_raw, 3), 4, P(d_raw), 4, n,
                                P(dfa), P(dfb), None, None, 1, None, None, 0, st), "bwd pair, inputs only")
    assert torch.equal(dfa, res[False][1]) and torch.equal(dfb, res[False][2])
    assert lib.us_mlp_pair_supported(A, ctypes.byref(us.make_mlp_desc(32, width, 3 - n_hidden, 3, "sigmoid", bias, prec))) == 0
    assert lib.us_mlp_pair_supported(ctypes.byref(us.make_mlp_desc(32, width, n_hidden, 1, "none", bias, "fp32")),
                                     ctypes.byref(us.make_mlp_desc(32, width, n_hidden, 3, "sigmoid", bias, "fp32"))) == 0
