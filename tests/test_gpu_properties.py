"""
GPU: properties of the two-grid encoder and its table gradient that hold at ANY size, checked at BASELINE.json's full sizes, where the
CPU oracle would take minutes: cfg2 (262 144 points, tables 2^16 / 2^19 at resolution 816) and cfg3 (786 432 points, 2^16 / 2^16 at 456).

  * conservation    the eight trilinear weights of a point sum to one, so per level and feature the table gradient sums to the sum of
                    dL/dfeature over the points (tcnn kernel_grid_backward reached from src/Mapper.py:444)
  * scaling         powers of two commute with every rounding on the path: encode(4 T) == 4 encode(T) and grad(4 dL) == 4 grad(dL),
                    bit for bit
  * linearity       encode(T1 + T2) == encode(T1) + encode(T2) up to fp32 rounding
  * order           the gradient does not depend on the order of the points (sums are formed in f64 and rounded once)
  * repeatability   US_GRID_BWD_DETERMINISTIC: two runs give the same bits
"""
import ctypes

import pytest
import torch

import unislam_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SIZES = [pytest.param(16, 19, 816, 4096 * 64, id="cfg2"), pytest.param(16, 16, 456, 8192 * 96, id="cfg3")]


def _cfg(log2T, res):
    return {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2T, "base_resolution": 16,
            "per_level_scale": O.per_level_scale(res)}


def _rays(n, S, seed):
    """S samples along each ray through the unit cube (some of them outside it: the encoder clamps)"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    R = n // S
    o = torch.rand((R, 1, 3), device=DEV, generator=g) * 0.8 + 0.1
    d = torch.randn((R, 1, 3), device=DEV, generator=g) * 0.3
    t = torch.linspace(0, 1, S, device=DEV).reshape(1, S, 1)
    return (o + d * t).reshape(-1, 3).contiguous(), g


@pytest.fixture(scope="module")
def us():
    import unislam_amd
    assert torch.cuda.is_available()
    return unislam_amd


class _Pair:
    def __init__(self, us, l2a, l2b, res, n, g):
        from unislam_amd import _lib as L
        self.L, self.lib, self.st, self.P = L, L.lib(), L.stream(), L.ptr
        self.ea, self.eb = us.HashGridEncoding(3, _cfg(l2a, res)).to(DEV), us.HashGridEncoding(3, _cfg(l2b, res)).to(DEV)
        self.ta = torch.randn(self.ea.params.shape, device=DEV, generator=g) * 0.2
        self.tb = torch.randn(self.eb.params.shape, device=DEV, generator=g) * 0.2
        self.da, self.db = ctypes.byref(self.ea.desc), ctypes.byref(self.eb.desc)
        self.n = n
        assert self.lib.us_hashgrid_joint_supported(self.da, self.db, n) == 1
        self.nbytes = int(self.lib.us_hashgrid_joint_workspace_bytes(self.da, self.db, n))
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=DEV)

    def encode(self, ta, tb, x):
        oa, ob = torch.empty((16, self.n, 2), device=DEV), torch.empty((16, self.n, 2), device=DEV)
        self.L.check(self.lib.us_hashgrid_fwd_joint(self.da, self.db, self.P(ta), self.P(tb), self.P(x), self.n, self.P(oa), self.P(ob), 3,
                                                    None, 0, self.st), "fwd joint")
        return oa, ob

    def grad(self, x, dya, dyb, extra=0):
        ga = torch.full((self.ea.desc.n_params,), 3.0, device=DEV)
        gb = torch.full((self.eb.desc.n_params,), -2.0, device=DEV)
        self.L.check(self.lib.us_hashgrid_bwd_joint(self.da, self.db, self.P(x), self.P(dya), self.P(dyb), self.n, self.P(ga), self.P(gb),
                                                    3 | self.L.US_GRID_BWD_OVERWRITE | extra, self.P(self.ws), self.nbytes, self.st), "bwd joint")
        return ga, gb


@pytest.mark.parametrize("l2a,l2b,res,n", SIZES)
def test_encoder_scaling_and_linearity_at_full_size(us, l2a, l2b, res, n):
    x, g = _rays(n, 64 if n == 4096 * 64 else 96, 11)
    p = _Pair(us, l2a, l2b, res, n, g)
    oa, ob = p.encode(p.ta, p.tb, x)
    assert torch.isfinite(oa).all() and torch.isfinite(ob).all()
    # powers of two commute with the roundings of the trilinear blend
    sa, sb = p.encode(p.ta * 4.0, p.tb * 0.25, x)
    assert torch.equal(sa, oa * 4.0) and torch.equal(sb, ob * 0.25)
    # linear in the table
    ua = torch.randn(p.ta.shape, device=DEV, generator=g) * 0.2
    ub = torch.randn(p.tb.shape, device=DEV, generator=g) * 0.2
    la, lb = p.encode(ua, ub, x)
    ca, cb = p.encode(p.ta + ua, p.tb + ub, x)
    for c, a, b in ((ca, oa, la), (cb, ob, lb)):
        assert float((c - (a + b)).abs().max()) <= 4e-6 * float(c.abs().max())
    # a constant table comes back as that constant (the weights of a point sum to one)
    ka, kb = p.encode(torch.full_like(p.ta, 0.375), torch.full_like(p.tb, -1.5), x)
    assert float((ka - 0.375).abs().max()) <= 1e-6 and float((kb + 1.5).abs().max()) <= 4e-6


@pytest.mark.parametrize("l2a,l2b,res,n", SIZES)
def test_table_gradient_properties_at_full_size(us, l2a, l2b, res, n):
    S = 64 if n == 4096 * 64 else 96
    x, g = _rays(n, S, 12)
    p = _Pair(us, l2a, l2b, res, n, g)
    dya, dyb = torch.randn((16, n, 2), device=DEV, generator=g), torch.randn((16, n, 2), device=DEV, generator=g)
    dya[:, ::9] = 0.0
    ga, gb = p.grad(x, dya, dyb)
    assert torch.isfinite(ga).all() and torch.isfinite(gb).all()
    # conservation, per level and feature (f64 sums on both sides; the bar is the rounding of the entries to fp32)
    for enc, gg, dy in ((p.ea, ga, dya), (p.eb, gb, dyb)):
        off = [int(enc.desc.offset[l]) for l in range(17)]
        for l in range(16):
            lvl = gg[2 * off[l]:2 * off[l + 1]].double().reshape(-1, 2)
            want, got = dy[l].double().sum(0), lvl.sum(0)
            bar = 1e-6 * float(lvl.abs().sum()) + 1e-9
            assert float((got - want).abs().max()) <= bar, (l, got.tolist(), want.tolist())
    # scaling by a power of two: the same bits, scaled
    sa, sb = p.grad(x, dya * 8.0, dyb * 0.5)
    assert torch.equal(sa, ga * 8.0) and torch.equal(sb, gb * 0.5)
    # the order of the points does not matter: rays shuffled, and every point shuffled (no run of samples inside a cell survives)
    for perm in (torch.randperm(n // S, device=DEV, generator=g).repeat_interleave(S) * S + torch.arange(S, device=DEV).repeat(n // S),
                 torch.randperm(n, device=DEV, generator=g)):
        pa, pb = p.grad(x[perm].contiguous(), dya[:, perm].contiguous(), dyb[:, perm].contiguous())
        for a, b in ((pa, ga), (pb, gb)):
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    # repeatable with the deterministic flag (no bin split over several workgroups' float atomics)
    d1 = p.grad(x, dya, dyb, p.L.US_GRID_BWD_DETERMINISTIC)
    d2 = p.grad(x, dya, dyb, p.L.US_GRID_BWD_DETERMINISTIC)
    assert torch.equal(d1[0], d2[0]) and torch.equal(d1[1], d2[1])
    for a, b in ((d1[0], ga), (d1[1], gb)):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
