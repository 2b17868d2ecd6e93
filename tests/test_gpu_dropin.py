"""
The reference's OWN call sequence (tcnn.Encoding / Decoders.forward / Renderer.render_batch_ray under torch autograd + an Adam with the
reference's three param groups: src/networks/decoders.py:91-105,182-205, src/utils/Renderer.py:59-152, src/Mapper.py:111-139,358-364,
443-445) on the kernels of the straight-line mapping step: Decoders.forward as ONE autograd node (joint encoder, decoder pair, joint
binned table gradient), one-launch sampling, and unislam_amd.optim.Adam (one launch per optimizer.step()).

Held against the two-module path (encoder and decoder as separate autograd nodes; r6: on the counted / scanned kernels with a cached
scratch): features' effect on raw within 1e-6, table gradients within 1e-4 of the largest gradient, decoder gradients 1e-4 -- HIP against
HIP.  What anchors the node itself: fixtures g4 / g7 / g9 (tests/test_gpu_parity.py, test_gpu_step.py) run THROUGH it since r5, and (r6)
test_one_node_forward_at_full_size_against_the_oracle / test_module_seam_alone_against_the_oracle below compare node and seam with the CPU
oracle directly at BASELINE configs[1]'s 262 144 points.
"""
import copy

import numpy as np
import pytest
import torch

import unislam_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BOUND = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])


@pytest.fixture(scope="module")
def us():
    import unislam_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return unislam_amd


def enc_cfg(log2T, res=816):
    return {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2T,
            "base_resolution": 16, "per_level_scale": O.per_level_scale(res)}


def _cfg(tcnn=False, prec="fp32", n_strat=32, n_imp=8, hidden=16):
    return {"rendering": {"perturb": True, "n_stratified": n_strat, "n_importance": n_imp, "learnable_beta": True}, "scale": 1,
            "grid_mode": "hash_grid", "grid": {"tcnn_network": tcnn}, "model": {"c_dim": 32, "truncation": 0.06, "mlp_precision": prec}}


def _model(us, tcnn, prec, log2T=(16, 19), hidden=16, n_blocks=2, seed=0):
    torch.manual_seed(seed)
    dec = us.Decoders(_cfg(tcnn, prec), c_dim=32, hidden_size=hidden, truncation=0.06, n_blocks=n_blocks).to(DEV)
    es, ec = us.HashGridEncoding(3, enc_cfg(log2T[0])).to(DEV), us.HashGridEncoding(3, enc_cfg(log2T[1])).to(DEV)
    with torch.no_grad():
        es.params.copy_(torch.randn_like(es.params) * 0.1); ec.params.copy_(torch.randn_like(ec.params) * 0.1)
    return dec, es, ec


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("tcnn,prec,n,with_x", [(False, "fp32", 70001, True), (False, "bf16", 4096 * 8, False), (True, "bf16", 20000, True),
                                                (True, "fp32", 333, False), (False, "fp32", 1, True)])
def test_one_node_decoders_forward_equals_two_module_path(us, tcnn, prec, n, with_x):
    """Decoders.forward through _DecodersFusedFn against the same module with fused = False (us_hashgrid_fwd / us_mlp_fwd per decoder,
    us_hashgrid_bwd_binned per table): raw, dL/dp, both table gradients, every decoder gradient."""
    dec, es, ec = _model(us, tcnn, prec, hidden=32 if prec == "bf16" else 16)
    torch.manual_seed(1)
    p = (torch.rand(n, 3, device=DEV) * 1.1 - 0.05)             # some coordinates outside [0,1]: the clamp's branch
    probe = torch.randn(n, 4, device=DEV)
    res = {}
    for fused in (True, False):
        dec.fused = fused
        for m in (dec, es, ec):
            m.zero_grad(set_to_none=True)
        x = p.clone().requires_grad_(with_x)
        raw = dec(x, ([es], [ec]))
        assert raw.shape == (n, 4)
        (raw * probe).sum().backward()
        res[fused] = dict(raw=raw.detach().clone(), gx=None if not with_x else x.grad.clone(), gs=es.params.grad.clone(), gc=ec.params.grad.clone(),
                          gd={k: v.grad.clone() for k, v in dec.named_parameters() if v.grad is not None})
    a, b = res[True], res[False]
    tol_raw = 1e-6 if prec == "fp32" else 1e-6            # same arithmetic per element in both paths: only the feature layout differs
    assert _rel(a["raw"], b["raw"]) <= tol_raw
    assert _rel(a["gs"], b["gs"]) <= 1e-4 and _rel(a["gc"], b["gc"]) <= 1e-4
    if with_x:
        assert _rel(a["gx"], b["gx"]) <= 1e-4
    assert set(a["gd"]) == set(b["gd"]) and len(a["gd"]) >= (2 if tcnn else 12)
    for k in a["gd"]:
        assert _rel(a["gd"][k], b["gd"][k]) <= 1e-4, k


def test_one_node_forward_without_gradients_and_reshape(us):
    """under no_grad (render_img, the mesher) the node runs the encoder without the table gradient's bookkeeping; [R,S,3] in, [R,S,4] out"""
    dec, es, ec = _model(us, False, "fp32")
    p = torch.rand(37, 40, 3, device=DEV)
    with torch.no_grad():
        a = dec(p, ([es], [ec]))
        dec.fused = False
        b = dec(p, ([es], [ec]))
    assert a.shape == (37, 40, 4) and _rel(a, b) <= 1e-6
    assert dec._fused_ws is None                              # no scratch was made


def test_overtaken_forward_pass_counts_again(us):
    """two forward passes before the first backward pass: the cached binning counts belong to the second; the first must not use them"""
    dec, es, ec = _model(us, False, "fp32")
    torch.manual_seed(2)
    pa, pb = torch.rand(30000, 3, device=DEV), torch.rand(50000, 3, device=DEV)
    qa, qb = torch.randn(30000, 4, device=DEV), torch.randn(50000, 4, device=DEV)
    ra = dec(pa, ([es], [ec])); rb = dec(pb, ([es], [ec]))
    (ra * qa).sum().backward()                                  # overtaken by rb's forward pass
    ga = es.params.grad.clone(), ec.params.grad.clone()
    es.zero_grad(); ec.zero_grad()
    (rb * qb).sum().backward()
    gb = es.params.grad.clone(), ec.params.grad.clone()
    dec.fused = False
    for (p, q, g) in ((pa, qa, ga), (pb, qb, gb)):
        for m in (dec, es, ec):
            m.zero_grad(set_to_none=True)
        (dec(p, ([es], [ec])) * q).sum().backward()
        assert _rel(g[0], es.params.grad) <= 1e-4 and _rel(g[1], ec.params.grad) <= 1e-4


def test_packed_parameters_follow_the_optimiser_state_dict_and_device_moves(us):
    """the nn.Linear parameters become views of one packed vector: optimizer steps, load_state_dict, deepcopy and .to() keep the
    kernel's vector and the module's parameters one thing"""
    dec, es, ec = _model(us, False, "fp32")
    p = torch.rand(5000, 3, device=DEV)
    sr = ([es], [ec])
    r0 = dec(p, sr).detach().clone()
    base = dec.linears[0].weight.data_ptr()
    assert dec.c_output_linear.bias.data_ptr() > base and dec._flat is not None
    # an in-place update through the parameter (what torch.optim does) is seen by the kernels
    with torch.no_grad():
        dec.linears[0].weight.mul_(0.5); dec.c_output_linear.bias.add_(0.25)
    r1 = dec(p, sr).detach().clone()
    dec.fused = False
    assert _rel(r1, dec(p, sr)) <= 1e-6 and _rel(r1, r0) > 1e-3
    dec.fused = True
    assert dec.linears[0].weight.data_ptr() == base            # no re-packing happened
    # state_dict round trip (Tracker.py:254: self.decoders.load_state_dict(self.shared_decoders.state_dict()))
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    with torch.no_grad():
        for q in dec.parameters():
            q.add_(1.0)
    dec.load_state_dict(sd)
    assert _rel(dec(p, sr), r1) <= 1e-6
    d2 = copy.deepcopy(dec)                                     # Tracker.py:106
    assert _rel(d2(p, sr), r1) <= 1e-6
    d3 = dec.cpu().to(DEV)                                      # new storage per parameter: packed again on the next call
    assert _rel(d3(p, sr), r1) <= 1e-6


def test_optim_adam_equals_torch_adam(us):
    """unislam_amd.optim.Adam against torch.optim.Adam: three param groups with their own learning rates set after construction
    (src/Mapper.py:118-126), tensors of odd lengths and unaligned views, a parameter without a gradient, five steps"""
    g = torch.Generator().manual_seed(5)
    shapes = [(16, 32), (16,), (1, 16), (1,), (100003,), (1 << 20,), (7, 3)]
    flat = torch.randn(40, generator=g).to(DEV)
    def make():
        ps = [torch.randn(*s, generator=torch.Generator().manual_seed(i)).to(DEV).requires_grad_(True) for i, s in enumerate(shapes)]
        odd = flat.clone()[1:22].view(7, 3)                   # a view at a 4-byte offset: not 16-byte aligned
        ps[-1] = odd.detach().requires_grad_(True)
        return ps
    pa, pb = make(), make()
    groups = lambda ps: [{"params": ps[:4], "lr": 0}, {"params": [ps[4]], "lr": 0}, {"params": ps[5:] , "lr": 0}]
    oa, ob = torch.optim.Adam(groups(pa), foreach=False), us.optim.Adam(groups(pb))
    for o in (oa, ob):
        o.param_groups[0]["lr"], o.param_groups[1]["lr"], o.param_groups[2]["lr"] = 1e-3, 0.05, 0.02
    for step in range(5):
        for k, (a, b) in enumerate(zip(pa, pb)):
            if k == 3 and step < 2:
                a.grad = b.grad = None                          # joins later: its own step count
                continue
            gr = torch.randn(a.shape, generator=g).to(DEV) * (0.0 if (step == 2 and k == 4) else 0.1)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    assert int(ob.state[pb[3]]["step"]) == 3 and int(ob.state[pb[0]]["step"]) == 5


def _reference_shaped_iteration(us, rend, dec, es, ec, opt, pools, c2ws, n_per, w, t_rand=None, idx=None):
    """src/Mapper.py:372-445 in the reference's own calls"""
    depths, colors, dirs = pools
    H = W = 0
    opt.zero_grad()
    ro, rd, gd, gc = us.common.get_samples_all(0, H, 0, W, n_per, H, W, 0, 0, 0, 0, c2ws, depths, colors, DEV, dirs, indices=idx)
    inside = us.common.bbox_filter(ro, rd, gd, BOUND)          # Mapper.py:396-406: the pre-filter and its four compactions
    ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
    if t_rand is not None:
        t_rand = t_rand[inside]
    ret = rend.render_batch_ray(([es], [ec]), dec, rd, ro, DEV, 0.06, gt_depth=gd, t_rand=t_rand)
    loss = us.mapping_loss(ret, gd, gc, 0.06, w)
    loss.backward()
    opt.step()
    return loss, ret, inside


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_reference_shaped_iteration_equals_mapstep(us, prec):
    """three mapping iterations through get_samples_all -> Renderer.render_batch_ray -> mapping_loss -> backward -> Adam.step() against
    MapStep.iterate on the same rays, jitter and initial model: losses, rendered depth / colour, and the model after the three steps"""
    import types
    torch.manual_seed(0)
    hidden = 32 if prec == "bf16" else 16
    dec, es, ec = _model(us, False, prec, hidden=hidden)
    dec2, es2, ec2 = copy.deepcopy(dec), copy.deepcopy(es), copy.deepcopy(ec)
    cfg = _cfg(False, prec)
    rend = us.Renderer(cfg, types.SimpleNamespace(bound=BOUND, device=DEV, H=12, W=16, fx=10., fy=10., cx=7.5, cy=5.5))
    w = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
    lr = dict(decoders=1e-3, sdf_grid=0.05, color_grid=0.05)
    b, P, n_per = 4, 5000, 512
    g = torch.Generator().manual_seed(3)
    c2ws = torch.eye(4).repeat(b, 1, 1)
    c2ws[:, :3, 3] = torch.tensor([3.0, 1.2, 0.0]) + torch.randn(b, 3, generator=g) * 0.1
    dirs = torch.randn(b, P, 3, generator=g); dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    pools = ((torch.rand(b, P, generator=g) * 2 + 0.4).to(DEV), torch.rand(b, P, 3, generator=g).to(DEV), dirs.to(DEV))
    c2ws = c2ws.to(DEV)
    opt = us.optim.Adam([{"params": list(dec.parameters()), "lr": lr["decoders"]}, {"params": [es.params], "lr": lr["sdf_grid"]},
                         {"params": [ec.params], "lr": lr["color_grid"]}])
    step = us.MapStep(es2, ec2, dec2, BOUND, 32, 8, 0.06, w, lr, max_rays=b * n_per)
    for it in range(3):
        idx = torch.randint(P, (b, n_per), generator=g).to(DEV)
        t_rand = torch.rand(b * n_per, 40, generator=g).to(DEV)
        loss, ret, inside = _reference_shaped_iteration(us, rend, dec, es, ec, opt, pools, c2ws, n_per, w, t_rand, idx)
        ro, rd, gd, gc = us.common.get_samples_all(0, 0, 0, 0, n_per, 0, 0, 0, 0, 0, 0, c2ws, *pools[:2], DEV, pools[2], indices=idx)
        loss2 = step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
        r2 = step.rendered()
        assert abs(float(loss) - float(loss2)) <= 2e-5 * abs(float(loss2)), (it, float(loss), float(loss2))
        assert 0 < int(inside.sum()) < inside.numel()            # some rays leave the box before their depth: both paths drop them
        assert _rel(ret[2], r2[2][inside]) <= 1e-4 and _rel(ret[3], r2[3][inside]) <= 1e-4
    # tables after Adam steps: an entry whose gradient is a near-cancelling sum takes a step that depends on the order of a float sum (Adam divides
    # by the entry's own |g|): all but <= 0.1 % of the entries within 2e-4 of the largest, those within 2e-3 (lr = 0.05)
    for a_, b_ in ((es.params, es2.params), (ec.params, ec2.params)):
        d = (a_.detach() - b_.detach()).abs()
        scale = float(b_.detach().abs().max())
        assert float((d > 2e-4 * scale).float().mean()) <= 1e-3 and float(d.max()) <= 2e-3, (float((d > 2e-4 * scale).float().mean()), float(d.max()))
    for (k, a), (_, b_) in zip(dec.named_parameters(), dec2.named_parameters()):
        assert _rel(a, b_) <= 2e-4, k


# ---------------------------------------------------------------------------------------------- r6: the node against the ORACLE, the seam alone
def _oracle_model(dec, es, ec, log2T):
    """the CPU oracle with this model's parameters (torch-MLP decoders: the layout fixture g4 pins; the C hash grid)"""
    od = O.DecodersOracle()
    od.load_state_dict({k: v.detach().cpu() for k, v in dec.state_dict().items()})
    oes, oec = O.HashGridOracle(3, enc_cfg(log2T[0])), O.HashGridOracle(3, enc_cfg(log2T[1]))
    with torch.no_grad():
        oes.params.copy_(es.params.detach().cpu()); oec.params.copy_(ec.params.detach().cpu())
    return od, oes, oec


def test_one_node_forward_at_full_size_against_the_oracle(us):
    """BASELINE configs[1]'s 4096 x 64 = 262 144 points through _DecodersFusedFn (fp32 decoders) against the CPU oracle on the same
    parameters: raw 1e-6, both table gradients 1e-4 of the largest entry, decoder gradients 1e-4, dL/dp 1e-3 -- this file's other tests
    compare HIP with HIP (one node against two modules); this one anchors the node itself (g4 / g7 / g9 run through it at fixture sizes)."""
    log2T = (16, 19)
    dec, es, ec = _model(us, False, "fp32", log2T=log2T)
    od, oes, oec = _oracle_model(dec, es, ec, log2T)
    n = 4096 * 64
    torch.manual_seed(3)
    p = torch.rand(n, 3) * 1.04 - 0.02
    probe = torch.randn(n, 4)
    x = p.to(DEV).requires_grad_(True)
    raw = dec(x, ([es], [ec]))
    (raw * probe.to(DEV)).sum().backward()
    xo = p.clone().requires_grad_(True)
    raw_o = od(xo, ([oes], [oec]))
    (raw_o * probe).sum().backward()
    np.testing.assert_allclose(raw.detach().cpu().numpy(), raw_o.detach().numpy(), rtol=0, atol=1e-6)
    assert _rel(es.params.grad, oes.params.grad) <= 1e-4 and _rel(ec.params.grad, oec.params.grad) <= 1e-4
    n_dec = 0
    for (k, a), (_, b) in zip(dec.named_parameters(), od.named_parameters()):
        assert (a.grad is None) == (b.grad is None), k           # (beta takes no part in Decoders.forward)
        if a.grad is not None:
            assert _rel(a.grad, b.grad) <= 1e-4, k
            n_dec += 1
    assert n_dec == 12
    # dL/dp jumps where a coordinate sits on a cell face of a fine level (and is zero outside [0,1]): compare away from the largest 0.1 %
    d = (x.grad.cpu() - xo.grad).abs().reshape(-1)
    assert float(torch.quantile(d[::7], 0.999)) <= 1e-3 * float(xo.grad.abs().max())


@pytest.mark.parametrize("n,det", [(20000, False), (4096 * 16, True), (300, False)])
def test_module_seam_alone_against_the_oracle(us, n, det):
    """`import unislam_amd.tcnn as tcnn` ALONE: enc(x) as its own autograd node -- r6: us_hashgrid_fwd_counted +
    us_hashgrid_bwd_scan in the forward pass, us_hashgrid_bwd_binned(COUNTED | SCANNED) in the backward pass, scratch cached on the module
    (n >= 16384; below that the plain encoder + the sliced gradient) -- and net(h), against the CPU oracle: features 1e-6, table gradient 1e-4,
    input gradient 1e-3 away from cell faces.  The scratch is allocated once."""
    import unislam_amd.tcnn as tcnn
    torch.manual_seed(4)
    enc = tcnn.Encoding(n_input_dims=3, encoding_config=enc_cfg(17), dtype=torch.float).to(DEV)
    if det:
        enc.grid_bwd_flags = us._lib.US_GRID_BWD_DETERMINISTIC
    oenc = O.HashGridOracle(3, enc_cfg(17))
    with torch.no_grad():
        enc.params.copy_(torch.randn_like(enc.params) * 0.1); oenc.params.copy_(enc.params.detach().cpu())
    grads = []
    for rep in range(2):
        p = torch.rand(n, 3)
        probe = torch.randn(n, 32)
        x = p.to(DEV).requires_grad_(True)
        enc.zero_grad(set_to_none=True); oenc.zero_grad(set_to_none=True)
        h = enc(x)
        (h * probe.to(DEV)).sum().backward()
        xo = p.clone().requires_grad_(True)
        ho = oenc(xo)
        (ho * probe).sum().backward()
        np.testing.assert_allclose(h.detach().cpu().numpy(), ho.detach().numpy(), rtol=0, atol=1e-6)
        assert _rel(enc.params.grad, oenc.params.grad) <= 1e-4
        d = (x.grad.cpu() - xo.grad).abs().reshape(-1)
        assert float(torch.quantile(d, 0.999)) <= 1e-3 * float(xo.grad.abs().max())
        grads.append(enc.params.grad.clone())
        ws = enc._ws
        assert (ws is not None) == (n >= 16384)
    assert enc._ws is ws                                        # one scratch for both iterations
    # a forward pass overtaken by another one of the same module counts again in a scratch of its own
    pa, pb = torch.rand(n, 3, device=DEV), torch.rand(n, 3, device=DEV)
    qa = torch.randn(n, 32, device=DEV)
    enc.zero_grad(set_to_none=True)
    ha = enc(pa); enc(pb)
    (ha * qa).sum().backward()
    g_overtaken = enc.params.grad.clone()
    enc.zero_grad(set_to_none=True)
    (enc(pa) * qa).sum().backward()
    assert _rel(g_overtaken, enc.params.grad) <= 1e-5
    # the network as its own node: the scratch of its backward pass is cached too
    net = tcnn.Network(n_input_dims=32, n_output_dims=3, network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                                                                        "n_neurons": 16, "n_hidden_layers": 1}).to(DEV)
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        net(enc(pa)).sum().backward()
        w = net._ws
    assert net._ws is w and net.params.grad is not None


def test_backward_after_an_in_place_update_of_the_decoders_raises(us):
    """the node's backward pass reads the decoders' weights from the packed vector as it is THEN; an optimizer.step() between forward and
    backward (torch.optim.Adam or unislam_amd.optim.Adam) must raise like a modified saved tensor does, not differentiate at the new weights"""
    for kind in ("torch", "fused"):
        dec, es, ec = _model(us, False, "fp32")
        opt = (us.optim.Adam if kind == "fused" else torch.optim.Adam)(list(dec.parameters()), lr=1e-3)   # (the tables ARE saved tensors: torch checks those itself)
        p = torch.rand(20000, 3, device=DEV)
        raw = dec(p, ([es], [ec]))
        raw.sum().backward(retain_graph=True)                   # fine: nothing changed yet
        opt.step()
        with pytest.raises(us.UniSlamHipError, match="modified in place"):
            raw.sum().backward()
        dec.zero_grad(set_to_none=True)
        dec(p, ([es], [ec])).sum().backward()                   # a fresh forward pass is fine again
