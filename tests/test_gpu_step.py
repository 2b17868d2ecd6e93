"""
GPU: MapStep (the straight-line, autograd-free mapping iteration on preallocated buffers) against the autograd path
(Renderer + Decoders + losses + torch.optim.Adam) and against the CPU oracle, on identical rays and random draws.
"""
import copy
import os
import types

import numpy as np
import pytest
import torch

import unislam_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BOUND = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])
W = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
LR = dict(decoders=0.001, sdf_grid=0.05, color_grid=0.05)


def _cfg(tcnn, ns=32, ni=8):
    return {"rendering": {"perturb": True, "n_stratified": ns, "n_importance": ni}, "scale": 1, "grid_mode": "hash_grid",
            "grid": {"tcnn_network": tcnn}}


def _ecfg(log2T):
    return {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2T,
            "base_resolution": 16, "per_level_scale": O.per_level_scale(816)}


def _scene(us, tcnn, seed=0):
    torch.manual_seed(seed)
    dec = us.Decoders(_cfg(tcnn), c_dim=32, truncation=0.06).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
    with torch.no_grad():
        es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
    return dec, es, ec


def _rays(R, seed=1, zero_depth=False, outside=False):
    g = torch.Generator().manual_seed(seed)
    ro = torch.tensor([[3.0, 1.2, 0.0]]).repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05
    rd = torch.randn(R, 3, generator=g); rd = rd / rd.norm(dim=-1, keepdim=True)
    gd = torch.rand(R, generator=g) * 2 + 0.4
    if zero_depth:
        gd[::5] = 0.0
    if outside:
        gd[1::7] = 50.0                       # beyond the scene box -> dropped by the pre-filter
    gc = torch.rand(R, 3, generator=g)
    return ro.to(DEV), rd.to(DEV), gd.to(DEV), gc.to(DEV)


@pytest.mark.parametrize("tcnn", [False, True])
@pytest.mark.parametrize("outside", [False, True])
def test_mapstep_matches_autograd_path(tcnn, outside):
    import unislam_amd as us
    dec, es, ec = _scene(us, tcnn)
    dec2, es2, ec2 = copy.deepcopy(dec), copy.deepcopy(es), copy.deepcopy(ec)
    R, S = 300, 40
    ro, rd, gd, gc = _rays(R, outside=outside)
    t_rand = torch.rand(R, S, device=DEV)
    # --- autograd path, reference structure (filter -> render -> loss -> backward -> Adam)
    rend = us.Renderer(_cfg(tcnn), types.SimpleNamespace(bound=BOUND, device=DEV, H=12, W=16, fx=10., fy=10., cx=7.5, cy=5.5))
    opt = torch.optim.Adam([{"params": list(dec2.parameters()), "lr": LR["decoders"]},
                            {"params": [es2.params], "lr": LR["sdf_grid"]}, {"params": [ec2.params], "lr": LR["color_grid"]}])
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
    for it in range(3):
        inside = us.common.bbox_filter(ro, rd, gd, BOUND)
        ret = rend.render_batch_ray(([es2], [ec2]), dec2, rd[inside], ro[inside], DEV, 0.06, gt_depth=gd[inside], t_rand=t_rand[inside])
        loss_a = us.mapping_loss(ret, gd[inside], gc[inside], 0.06, W)
        opt.zero_grad(); loss_a.backward()
        # --- MapStep on the same inputs
        loss_b = step.forward_backward(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
        np.testing.assert_allclose(float(loss_b), float(loss_a), rtol=2e-5)
        ga = torch.cat([p.grad.reshape(-1) for p in (es2.params, ec2.params)])
        gb = torch.cat([es.params.grad.reshape(-1), ec.params.grad.reshape(-1)])
        assert torch.allclose(ga, gb, rtol=2e-4, atol=1e-5 * ga.abs().max().item())
        for (n, pa), (_, pb) in zip(dec2.named_parameters(), dec.named_parameters()):
            assert torch.allclose(pa.grad, pb.grad, rtol=2e-4, atol=1e-5 * max(1.0, pa.grad.abs().max().item())), n
        opt.step(); step.adam_step()
        for (n, pa), (_, pb) in zip(dec2.named_parameters(), dec.named_parameters()):
            assert torch.allclose(pa, pb, rtol=1e-4, atol=2e-6), n
        assert torch.allclose(es2.params, es.params, rtol=1e-4, atol=2e-5)
        assert torch.allclose(ec2.params, ec.params, rtol=1e-4, atol=2e-5)
    # the adopted modules still serialise with the reference's key names
    keys = set(dec.state_dict().keys())
    assert keys == ({"beta", "sdf_decoder.params", "color_decoder.params"} if tcnn else set(O.DecodersOracle().state_dict().keys()))


def test_mapstep_zero_depth_rays_and_oracle():
    import unislam_amd as us
    dec, es, ec = _scene(us, False, seed=3)
    R, S = 160, 40
    ro, rd, gd, gc = _rays(R, seed=4, zero_depth=True)
    far = O.bbox_far(ro.cpu(), rd.cpu(), BOUND)
    gd = torch.where(gd > 0, torch.minimum(gd, 0.9 * far.to(DEV)), gd)        # every ray inside the box: the oracle call below has no pre-filter
    n1, n0 = int((gd > 0).sum()), int((gd <= 0).sum())
    torch.manual_seed(9)
    tr1, tr0, u0 = torch.rand(n1, S), torch.rand(n0, 32), torch.rand(n0, 8)
    # oracle on CPU with the same draws
    od = O.DecodersOracle(); od.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    oes, oec = O.HashGridOracle(3, _ecfg(14)), O.HashGridOracle(3, _ecfg(15))
    with torch.no_grad():
        oes.params.copy_(es.params.cpu()); oec.params.copy_(ec.params.cpu())
    ret_o = O.render_batch_ray(([oes], [oec]), od, rd.cpu(), ro.cpu(), 0.06, gd.cpu(), BOUND, 32, 8, True,
                               {"z": tr1, "z_uni": tr0, "u": u0})
    loss_o = O.mapping_loss(ret_o, gd.cpu(), gc.cpu(), 0.06, W)
    loss_o.backward()
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=64)        # forces a buffer re-allocation too
    t_rand = torch.zeros(R, S); t_rand[(gd > 0).cpu()] = tr1
    loss = step.forward_backward(ro, rd, gd, gc, t_rand=t_rand.to(DEV), zero_depth_draws=(tr0.to(DEV), u0.to(DEV)))
    # the branch with its own draws (in-kernel generator): sorted samples inside the box for every zero-depth ray
    step.forward(ro, rd, gd, gc)
    z0 = step.rendered()[5][gd <= 0]
    assert bool((z0[:, 1:] >= z0[:, :-1]).all()) and bool((z0 >= 0).all()) and bool((z0.max(1)[0] <= far.to(DEV)[gd <= 0] + 0.011).all())
    loss = step.forward_backward(ro, rd, gd, gc, t_rand=t_rand.to(DEV), zero_depth_draws=(tr0.to(DEV), u0.to(DEV)))
    z = step.rendered()[5]
    np.testing.assert_allclose(z.cpu().numpy(), ret_o[5].numpy(), rtol=1e-4, atol=1e-5)
    # mask counts must agree before the loss can (a ray on the 0.99 opacity threshold would flip a count)
    _, unc_o, depth_o = ret_o[0], ret_o[1], ret_o[2]
    m_o = (gd.cpu() > 0) & ((1 - unc_o.detach()) > 0.99)
    print("counts hip", step.stats[5:].tolist(), "oracle depth-mask rays", int(m_o.sum()), "loss", float(loss), float(loss_o))
    np.testing.assert_allclose(step.rendered()[1].cpu().numpy(), unc_o.detach().numpy(), rtol=1e-3, atol=1e-6)
    assert int(step.stats[9]) == int(m_o.sum())
    np.testing.assert_allclose(float(loss), float(loss_o), rtol=1e-3)
    go = oes.params.grad
    assert torch.allclose(es.params.grad.cpu(), go, rtol=2e-3, atol=1e-4 * go.abs().max().item())


def test_mapstep_bench_shape_runs_and_decreases_loss():
    """BASELINE cfg2 shape: 4096 rays x 64 samples, room0 tables (log2T 16 / 19), 2x32 MLP; 20 iterations."""
    import unislam_amd as us
    torch.manual_seed(0)
    cfg = _cfg(False, 48, 16)
    dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(16)).to(DEV), us.HashGridEncoding(3, _ecfg(19)).to(DEV)
    step = us.MapStep(es, ec, dec, BOUND, 48, 16, 0.06, W, LR, max_rays=4096)
    ro, rd, gd, gc = _rays(4096, seed=2)
    losses = [float(step.iterate(ro, rd, gd, gc, has_zero_depth=False)) for _ in range(20)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("prec", ["fp32", "bf16", "f16"])
def test_mapstep_full_size_against_oracle(prec):
    """BASELINE cfg2 at its full size -- 4096 rays x 64 samples (48 + 16), room0 tables (log2T 16 / 19, res 816), 2 x 32 MLP -- one
    mapping iteration against the CPU oracle on the same rays and jitter: rendered depth / colour within 1e-3 relative (the
    north-star bound), loss, table and decoder gradients.  prec = bf16: the bench's headline decoders (bf16 MFMA, split-operand forward
    products): the same bounds on everything rendered and on the loss; the gradients, whose products see bf16 operands, norm-wise 2e-2."""
    import unislam_amd as us
    torch.manual_seed(5)
    cfg = dict(_cfg(False, 48, 16), model={"mlp_precision": prec})
    dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(16)).to(DEV), us.HashGridEncoding(3, _ecfg(19)).to(DEV)
    with torch.no_grad():                                       # features large enough for a non-trivial surface
        es.params.copy_(torch.randn(es.params.shape) * 0.2); ec.params.copy_(torch.randn(ec.params.shape) * 0.2)
    R, S = 4096, 64
    ro, rd, gd, gc = _rays(R, seed=6)
    far = O.bbox_far(ro.cpu(), rd.cpu(), BOUND)
    gd = torch.minimum(gd, 0.9 * far.to(DEV))                    # every ray inside the box: the oracle call has no pre-filter
    t_rand = torch.rand(R, S)
    od = O.DecodersOracle(hidden_size=32, n_blocks=2); od.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    oes, oec = O.HashGridOracle(3, _ecfg(16)), O.HashGridOracle(3, _ecfg(19))
    with torch.no_grad():
        oes.params.copy_(es.params.cpu()); oec.params.copy_(ec.params.cpu())
    ret_o = O.render_batch_ray(([oes], [oec]), od, rd.cpu(), ro.cpu(), 0.06, gd.cpu(), BOUND, 48, 16, True, {"z": t_rand})
    loss_o = O.mapping_loss(ret_o, gd.cpu(), gc.cpu(), 0.06, W)
    loss_o.backward()
    step = us.MapStep(es, ec, dec, BOUND, 48, 16, 0.06, W, LR, max_rays=R)
    loss = step.forward_backward(ro, rd, gd, gc, t_rand=t_rand.to(DEV), has_zero_depth=False)
    term, unc, depth, rgb = [t.cpu() for t in step.rendered()[:4]]
    if prec == "f16":
        # one f16 product per layer (the reference's tcnn arithmetic): inside the north star's 1e-3 norm-wise with a factor 5 to spare, the
        # worst single ray at 2e-3 -- which is why the split-operand bf16 kernels, not these, are the bench's decoders
        for a, b in ((depth, ret_o[2].detach()), (rgb, ret_o[3].detach())):
            assert float((a - b).norm() / b.norm()) < 3e-4
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=5e-3, atol=1e-5)
    else:
        np.testing.assert_allclose(depth.numpy(), ret_o[2].detach().numpy(), rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(rgb.numpy(), ret_o[3].detach().numpy(), rtol=1e-3, atol=1e-5)
    m_o = (gd.cpu() > 0) & ((1 - ret_o[1].detach()) > 0.99)
    assert abs(int(step.stats[9]) - int(m_o.sum())) <= 2          # rays sitting on the 0.99 opacity threshold may flip
    np.testing.assert_allclose(float(loss), float(loss_o), rtol=1e-3)
    for name, g_hip, g_o in (("sdf", es.params.grad.cpu(), oes.params.grad), ("colour", ec.params.grad.cpu(), oec.params.grad)):
        scale = g_o.abs().max().item()
        if prec == "fp32":
            assert torch.allclose(g_hip, g_o, rtol=5e-3, atol=2e-4 * scale), (name, float((g_hip - g_o).abs().max()), scale)
        assert float((g_hip - g_o).norm() / g_o.norm()) < (1e-3 if prec == "fp32" else 2e-2), (name, float((g_hip - g_o).norm() / g_o.norm()))
    for (n, pa), (_, pb) in zip(od.named_parameters(), dec.named_parameters()):
        if prec == "fp32":
            assert torch.allclose(pb.grad.cpu(), pa.grad, rtol=5e-3, atol=2e-4 * max(1e-3, pa.grad.abs().max().item())), n
        else:
            assert float((pb.grad.cpu() - pa.grad).norm() / pa.grad.norm()) < 2e-2, n


@pytest.mark.parametrize("mode", ["original", "no_mask"])
def test_trackstep_reproduces_reference_tracking_iteration(golden, mode):
    """TrackStep.iterate against the fixture captured from the reference's Tracker.optimize_tracking (g8)."""
    import unislam_amd as us
    g = golden("g8_tracking")
    T = torch.from_numpy
    H, Wd, fx, fy, cx, cy = g["intr"]; H, Wd = int(H), int(Wd)
    eh, ew = int(g["edge"][0]), int(g["edge"][1])
    cfg = _cfg(False)
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06)
    dec.load_state_dict({k[len("dec__"):].replace("__", "."): T(v) for k, v in g.items() if k.startswith("dec__")})
    dec = dec.to(DEV)
    ecfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 16,
            "per_level_scale": O.per_level_scale(64)}
    es, ec = us.HashGridEncoding(3, ecfg).to(DEV), us.HashGridEncoding(3, ecfg).to(DEV)
    with torch.no_grad():
        es.params.copy_(T(g["grid_s"])); ec.params.copy_(T(g["grid_c"]))
    pose = T(g["pose"]).to(DEV).clone().requires_grad_(True)
    opt = torch.optim.SGD([pose], lr=0.0)
    ts = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, dict(fs=10, center=200, tail=50, color=5, depth=1), mask_mode=mode, max_rays=16)
    torch.manual_seed(int(g["seed"]))
    n = int(g["n"])
    idx = torch.randint((H - 2 * eh) * (Wd - 2 * ew), (n,))                 # the reference's CPU draw
    # the reference jitters only the rays that pass the pre-filter; replay its draw onto those rows
    c2w = us.common.cam_pose_to_matrix(pose.detach())
    wi = Wd - 2 * ew
    i = (ew + idx % wi).float()[None].to(DEV); j = (eh + idx // wi).float()[None].to(DEV)
    ro, rd = us.common.get_rays_from_uv(i, j, c2w, H, Wd, fx, fy, cx, cy, DEV)
    gd = T(g["gt_depth"])[0, eh:H - eh, ew:Wd - ew].reshape(-1)[idx].to(DEV)
    inside = us.common.bbox_filter(ro.reshape(-1, 3), rd.reshape(-1, 3), gd, BOUND, require_depth=True)
    t_rand = torch.zeros(n, 40)
    t_rand[inside.cpu()] = torch.rand(int(inside.sum()), 40)
    loss, unc, valid = ts.iterate(pose, T(g["gt_color"]).to(DEV), T(g["gt_depth"]).to(DEV), n, opt, H, Wd, fx, fy, cx, cy, eh, ew,
                                  t_rand=t_rand.to(DEV), indices=idx.to(DEV))
    assert torch.equal(valid.bool(), inside)
    np.testing.assert_allclose(float(loss), float(g[f"{mode}_loss"]), rtol=1e-4)
    np.testing.assert_allclose(unc[valid.bool()].cpu().numpy(), g[f"{mode}_unc"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(pose.grad.cpu().numpy(), g[f"{mode}_gpose"], rtol=5e-3, atol=5e-3)


def test_captured_tracking_iteration_equals_eager():
    """hipGraph replay of TrackStep.iterate (incl. pose Adam) gives the eager trajectory"""
    import unislam_amd as us
    dec, es, ec = _scene(us, False, seed=7)
    for p in dec.parameters():
        p.requires_grad_(False)
    H, Wd, fx, fy, cx, cy = 60, 80, 40.0, 40.0, 39.5, 29.5
    g = torch.Generator().manual_seed(2)
    gt_depth = (torch.rand(1, H, Wd, generator=g) * 1.5 + 0.5).to(DEV); gt_color = torch.rand(1, H, Wd, 3, generator=g).to(DEV)
    n = 256
    idx = torch.randint((H - 8) * (Wd - 8), (n,), generator=g).to(DEV)
    t_rand = torch.rand(n, 40, generator=g).to(DEV)
    w = dict(fs=10, center=200, tail=50, color=5, depth=1)
    poses = []
    for captured in (False, True):
        pose = torch.nn.Parameter(torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]], device=DEV))
        opt = torch.optim.Adam([pose], lr=1e-3, betas=(0.5, 0.999), capturable=True)
        ts = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
        fn = lambda: ts.iterate(pose, gt_color, gt_depth, n, opt, H, Wd, fx, fy, cx, cy, 4, 4, t_rand=t_rand, indices=idx)
        for _ in range(2):                                    # Adam's state must exist before a capture (else the capture
            fn()                                              # records its zero-initialisation and every replay repeats it)
        if captured:
            it = us.CapturedIteration(fn, warmup=0)           # capturing does not execute: 3 replays = 3 more iterations
            for _ in range(3):
                loss, _, _ = it.replay()
        else:
            for _ in range(3):
                loss, _, _ = fn()
        poses.append((pose.detach().clone(), float(loss)))
    assert torch.allclose(poses[0][0], poses[1][0], rtol=1e-5, atol=1e-6)
    assert abs(poses[0][1] - poses[1][1]) <= 1e-4 * abs(poses[0][1])
    assert not torch.allclose(poses[0][0], torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]], device=DEV))   # it moved


def test_mapstep_single_rank_process_group_matches_plain():
    """the data-parallel code path (stats all-reduce, segment-wise async gradient all-reduce over RCCL) with world_size 1"""
    import os
    import torch.distributed as dist
    import unislam_amd as us
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()      # a free port, not a fixed one
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        outs = []
        for group, kw in ((None, {}), (True, {}), (True, dict(sharded_adam=True)), (True, dict(grad_comm="bf16"))):
            dec, es, ec = _scene(us, False, seed=11)
            step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=256, group=group, **kw)
            ro, rd, gd, gc = _rays(256, seed=12)
            t_rand = torch.rand(256, 40, generator=torch.Generator().manual_seed(1)).to(DEV)
            for _ in range(3):
                loss = step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
            outs.append((step.flat.clone(), float(loss)))
        assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6) and abs(outs[0][1] - outs[1][1]) < 1e-5
        # reduce-scatter + Adam on the rank's shard + all-gather (in place, RCCL): the same parameters
        assert torch.allclose(outs[0][0], outs[2][0], rtol=1e-5, atol=1e-6) and abs(outs[0][1] - outs[2][1]) < 1e-5
        # bf16 gradient payload: close, not identical (Adam steps are bounded by lr, a tiny gradient may flip its sign)
        d = (outs[3][0] - outs[0][0]).norm() / (outs[0][0]).norm()
        assert float(d) < 2e-2
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,variant", [(2, "plain"), (2, "sharded"), (2, "bf16"), (4, "plain"), (4, "sharded")])
def test_mapstep_ranks_on_one_gpu(tmp_path, world, variant):
    """world_size 2 and 4 through the real kernels: the rank processes (gloo, all on cuda:0) run MapStep + dist.dp_iterate on their
    own rays -- global loss counts, segment-wise gradient reduction, (sharded) Adam -- and end with the parameters of ONE process that
    sees the concatenated batch; the replicas stay identical.  (A test box admits 6 processes on its card: 4 ranks + this one.)"""
    import socket
    import subprocess
    import sys
    import unislam_amd as us
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "dp")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_two_ranks.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), out, variant], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0].decode()[-2000:])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    rs = [torch.load(f"{out}.{r}") for r in range(world)]
    r0 = rs[0]
    assert all(torch.equal(r0["flat"], r["flat"]) for r in rs[1:])               # replicas bit-identical (bf16 payload included)
    assert all(r["step_dev"] == r0["step_dev"] for r in rs[1:])                  # device-side step counts in lock-step
    # one process, all slices
    dec, es, ec = _scene(us, False, seed=11)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=256 * world)
    parts = [_rays(256, seed=100 + r, outside=(r % 2 == 1)) for r in range(world)]
    ro, rd, gd, gc = (torch.cat([p[k] for p in parts]) for k in range(4))
    t_rand = torch.cat([torch.rand(256, 40, generator=torch.Generator().manual_seed(200 + r)) for r in range(world)]).to(DEV)
    losses = [float(step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)) for _ in range(3)]
    one = step.flat.detach().cpu()
    if variant == "bf16":
        assert float((r0["flat"] - one).norm() / one.norm()) < 2e-2
        np.testing.assert_allclose(r0["losses"][0], losses[0], rtol=1e-5)          # the first loss precedes any rounded gradient
    else:
        np.testing.assert_allclose(r0["losses"], losses, rtol=2e-4)
        # Adam divides by sqrt(v): an entry whose gradient is at rounding level may move by up to lr in either direction
        close = torch.isclose(r0["flat"], one, rtol=1e-4, atol=2e-5)
        assert float((~close).float().mean()) < 1e-3, float((~close).float().mean())
        assert float((r0["flat"] - one).norm() / one.norm()) < 1e-3


@pytest.mark.parametrize("world,variant", [(2, "window"), (2, "window_extra"), (2, "window_graph"), (2, "window_sharded"), (4, "window"),
                                           (4, "window_graph")])
def test_joint_opt_window_ranks_on_one_gpu(tmp_path, world, variant):
    """The reference's DEFAULT mapping iteration (joint_opt, src/Mapper.py:359-376,443-459,518) data-parallel through the real kernels:
    every rank process (gloo, all on cuda:0) owns the frames {f : f mod W == rank} of a 6-frame window with their poses and pose moments
    (MapWindow.sharded), exchanges loss statistics and model gradients, and steps its own poses.  Model AND poses end as in ONE process
    that renders the whole window; the replicas stay bit-identical; the oldest pose stays fixed.  window_extra: with the extra rays of the
    newest frames (src/Mapper.py:385-393); window_graph: the rank-local launches replayed as hipGraph segments between the collectives
    (graph.SegmentedGraph); window_sharded: reduce-scatter + sharded Adam + all-gather."""
    import socket
    import subprocess
    import sys
    import unislam_amd as us
    import test_gpu_window as TW
    from _dp_two_ranks import window_draws
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "dpw")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_two_ranks.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), out, variant], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0].decode()[-3000:])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    rs = [torch.load(f"{out}.{r}") for r in range(world)]
    r0 = rs[0]
    assert all(torch.equal(r0["flat"], r["flat"]) for r in rs[1:])               # model replicas bit-identical
    assert all(torch.equal(r0["c2ws_all"], r["c2ws_all"]) for r in rs[1:])       # every rank collects the same window
    assert all(r["step_dev"] == 3.0 for r in rs)
    if variant == "window_graph":
        assert all(r["segments"] >= 4 for r in rs)                               # forward | stats | ... colour | rest | finish ... Adam
    # one process, the whole window, the same draws and jitter
    B, P, n_per = 6, 300, 40
    extra = (4, 15) if variant == "window_extra" else None
    c2ws, depths, colors, dirs = TW._window(B, P, 31)
    dec, es, ec = _scene(us, False, seed=11)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=B * n_per + (60 if extra else 0))
    win = us.MapWindow(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, cam_lr=1e-3, extra=extra, has_zero_depth=False)
    losses = []
    for it in range(3):
        idx, idx2, tr, tr2 = window_draws(B, P, n_per, extra, 40, it)
        t_rand = tr.reshape(-1, 40) if not extra else torch.cat([tr.reshape(-1, 40), tr2.reshape(-1, 40)])
        losses.append(float(win.iterate(idx.to(DEV), idx2.to(DEV) if extra else None, t_rand=t_rand.to(DEV))))
    np.testing.assert_allclose(r0["losses"], losses, rtol=2e-4)
    one = step.flat.detach().cpu()
    close = torch.isclose(r0["flat"], one, rtol=1e-4, atol=2e-5)
    assert float((~close).float().mean()) < 1e-3, float((~close).float().mean())
    assert float((r0["flat"] - one).norm() / one.norm()) < 1e-3
    ref = win.c2ws().cpu()
    assert torch.equal(r0["c2ws_all"][0], c2ws[0])                               # "we fix the oldest c2w", src/Mapper.py:374
    # Adam on 7 numbers per pose, lr 1e-3: three steps move an entry by <= 3e-3; a gradient component at rounding level may take another sign
    assert float((r0["c2ws_all"] - ref).abs().max()) < 2e-4, float((r0["c2ws_all"] - ref).abs().max())
    assert float((ref[1:] - c2ws[1:]).abs().max()) > 1e-3                        # the poses did move
    from unislam_amd.dist import shard_frames
    for r in range(world):                                                       # a rank's own view = its frames of the gathered window
        assert torch.equal(rs[r]["c2ws"], r0["c2ws_all"][shard_frames(B, r, world)])


def test_sharded_adam_rank_without_elements_keeps_its_step_count():
    """MapStep.adam_step(ranges) on a range that holds no parameter (a rank that owns only padding) still advances the device-side
    step count: it is Adam's bias-correction step and the salt of the sampler's jitter, and has to agree on every rank."""
    import unislam_amd as us
    dec, es, ec = _scene(us, False, seed=3)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=64)
    before, flat0 = float(step.step_dev[0]), step.flat.clone()
    pad_lo = step.o_tab_s + es.desc.n_params                                     # alignment padding behind the sdf table
    if pad_lo < step.o_tab_c:
        step.adam_step(ranges=[(pad_lo, step.o_tab_c)])
    else:
        step.adam_step(ranges=[])
    torch.cuda.synchronize()
    assert float(step.step_dev[0]) == before + 1.0
    assert torch.equal(step.flat, flat0)


def test_pose_kernels_match_torch_autograd():
    """us_pose_rays / us_pose_grad (closed-form chain rule through R(q)) against torch autograd on the reference's ray construction"""
    import unislam_amd as us
    from unislam_amd import _lib as L
    g = torch.Generator().manual_seed(21)
    H, Wd, fx, fy, cx, cy, eh, ew, n = 60, 80, 41.0, 43.0, 39.5, 29.5, 4, 5, 500
    pose = torch.tensor([0.7, -0.3, 0.5, 0.2, 3.0, 1.2, 0.1], device=DEV)              # not unit length: exercises the 2/|q|^2 factor
    idx = torch.randint((H - 2 * eh) * (Wd - 2 * ew), (n,), generator=g).to(DEV)
    depth = torch.rand(H, Wd, generator=g).to(DEV); color = torch.rand(H, Wd, 3, generator=g).to(DEV)
    f = lambda *s: torch.empty(s, device=DEV)
    ro, rd, dirs, gd, gc, gp = f(n, 3), f(n, 3), f(n, 3), f(n), f(n, 3), f(7)
    L.check(L.lib().us_pose_rays(L.ptr(pose), L.ptr(idx), n, L.host_floats([fx, fy, cx, cy]), ew, eh, Wd - 2 * ew, L.ptr(depth), L.ptr(color), Wd,
                                 L.ptr(ro), L.ptr(rd), L.ptr(dirs), L.ptr(gd), L.ptr(gc), L.stream()), "us_pose_rays")
    pr = pose.clone().requires_grad_(True)
    wi = Wd - 2 * ew
    i = (ew + idx % wi).float()[None]; j = (eh + idx // wi).float()[None]
    o_ref, d_ref = us.common.get_rays_from_uv(i, j, us.common.cam_pose_to_matrix(pr[None]), H, Wd, fx, fy, cx, cy, DEV)
    assert torch.allclose(rd, d_ref.reshape(-1, 3), rtol=1e-5, atol=1e-6) and torch.allclose(ro, o_ref.reshape(-1, 3))
    assert torch.equal(gd, depth[eh:H - eh, ew:Wd - ew].reshape(-1)[idx]) and torch.equal(gc, color[eh:H - eh, ew:Wd - ew].reshape(-1, 3)[idx])
    g_o, g_d = torch.randn(n, 3, generator=g).to(DEV), torch.randn(n, 3, generator=g).to(DEV)
    torch.autograd.backward([o_ref.reshape(-1, 3), d_ref.reshape(-1, 3)], [g_o, g_d])
    L.check(L.lib().us_pose_grad(L.ptr(pose), L.ptr(g_o), L.ptr(g_d), L.ptr(dirs), n, L.ptr(gp), L.stream()), "us_pose_grad")
    assert torch.allclose(gp, pr.grad, rtol=1e-4, atol=1e-4 * pr.grad.abs().max().item())


def test_fused_tracking_matches_autograd_tracking():
    """TrackStep.iterate_fused (pose->rays, pose gradient and pose Adam as HIP kernels) vs TrackStep.iterate (torch autograd + torch Adam)"""
    import unislam_amd as us
    dec, es, ec = _scene(us, False, seed=9)
    for p in dec.parameters():
        p.requires_grad_(False)
    H, Wd, fx, fy, cx, cy = 60, 80, 40.0, 40.0, 39.5, 29.5
    g = torch.Generator().manual_seed(4)
    gt_depth = (torch.rand(1, H, Wd, generator=g) * 1.5 + 0.5).to(DEV); gt_depth[0, 20, 20:30] = 0.0
    gt_color = torch.rand(1, H, Wd, 3, generator=g).to(DEV)
    n, eh, ew = 300, 4, 5
    w = dict(fs=10, center=200, tail=50, color=5, depth=1)
    pose0 = torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]], device=DEV)
    draws = [(torch.randint((H - 2 * eh) * (Wd - 2 * ew), (n,), generator=g).to(DEV), torch.rand(n, 40, generator=g).to(DEV)) for _ in range(4)]
    # reference structure: quaternion and translation as two Adam groups (Tracker.py:324-329)
    quad = torch.nn.Parameter(pose0[:, :4].clone()); T = torch.nn.Parameter(pose0[:, 4:].clone())
    opt = torch.optim.Adam([{"params": [T], "lr": 2e-3, "betas": (0.5, 0.999)}, {"params": [quad], "lr": 1e-3, "betas": (0.5, 0.999)}])
    ts_a = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
    ts_b = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
    ts_b.begin_frame(pose0, gt_color[0], gt_depth[0], 2e-3, 1e-3, H, Wd, fx, fy, cx, cy, eh, ew)
    for k, (idx, tr) in enumerate(draws):
        la, ua, va = ts_a.iterate(torch.cat([quad, T], -1), gt_color, gt_depth, n, opt, H, Wd, fx, fy, cx, cy, eh, ew, t_rand=tr, indices=idx)
        lb, ub, vb = ts_b.iterate_fused(n, t_rand=tr, indices=idx)
        ga = torch.cat([quad.grad, T.grad], -1).flatten()
        pa = torch.cat([quad, T], -1).detach().flatten()
        if k == 0:
            # identical pose: same rays up to 1 ulp, same loss; the pose gradient only differs where an ulp moves a sample across
            # a grid-cell face (the encoding is continuous, its derivative is not): a per-cent level effect on a 300-ray batch
            assert torch.equal(va, vb)
            np.testing.assert_allclose(float(lb), float(la), rtol=1e-5)
            assert (ts_b.g_pose - ga).norm() <= 0.03 * ga.norm()
        # Adam's first steps are +-lr per coordinate whatever the gradient size: the two trajectories stay within a few lr
        assert (ts_b.pose - pa).abs().max().item() <= 2.5 * 2e-3 * (k + 1)
        np.testing.assert_allclose(float(lb), float(la), rtol=2e-2)


def test_mapstep_reproduces_reference_optimize_mapping(golden):
    """MapStep driven like Mapper.optimize_mapping (first frame, 2 iterations incl. Adam) against the reference fixture g9"""
    import unislam_amd as us
    g = golden("g9_mapping")
    T = torch.from_numpy
    H, Wd, fx, fy, cx, cy = g["intr"]; H, Wd = int(H), int(Wd)
    dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06)
    dec.load_state_dict({k[len("dec0__"):].replace("__", "."): T(v) for k, v in g.items() if k.startswith("dec0__")})
    dec = dec.to(DEV)
    ecfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 16,
            "per_level_scale": O.per_level_scale(64)}
    es, ec = us.HashGridEncoding(3, ecfg).to(DEV), us.HashGridEncoding(3, ecfg).to(DEV)
    with torch.no_grad():
        es.params.copy_(T(g["grid_s0"])); ec.params.copy_(T(g["grid_c0"]))
    gt_depth, gt_color, c2w = T(g["gt_depth"]), T(g["gt_color"]), T(g["c2w"])
    cam = O.get_camera_rays(H, Wd, fx, fy, cx, cy)
    torch.manual_seed(int(g["seed"]))                                         # the reference's CPU random stream
    idx = torch.randperm(H * Wd)[:int(H * Wd * 0.1)]
    pool_c, pool_d, pool_r = (gt_color.reshape(-1, 3)[idx][None].to(DEV), gt_depth.reshape(-1)[idx][None].to(DEV),
                              cam.reshape(-1, 3)[idx][None].to(DEV))
    n = int(g["pixels"])
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=n)
    step.reset_optimizer(float(g["lr_factor"]))                               # lr_first_factor on the first frame (Mapper.py:520-525)
    for _ in range(int(g["iters"])):
        indices = torch.randint(pool_d.shape[1], (n,)).reshape(1, -1)
        ro, rd, gd, gc = us.common.get_samples_all(0, H, 0, Wd, n, H, Wd, fx, fy, cx, cy, c2w[None].to(DEV), pool_d, pool_c, DEV, pool_r,
                                                   indices=indices.to(DEV))
        inside = us.common.bbox_filter(ro, rd, gd, BOUND).cpu()
        n_depth = int(((gd.cpu() > 0) & inside).sum()); n_zero = int(((gd.cpu() <= 0) & inside).sum())
        t_rand = torch.zeros(n, 40)
        t_rand[((gd.cpu() > 0) & inside)] = torch.rand(n_depth, 40)          # the reference jitters only the rays it kept
        draws = [torch.rand(n_zero, 32).to(DEV), torch.rand(n_zero, 8).to(DEV)] if n_zero else []
        assert bool(inside[(gd.cpu() <= 0)].all())                            # zero-depth rays always pass the pre-filter
        real = torch.rand
        try:
            if draws:
                torch.rand = lambda *a, **k: draws.pop(0)
            step.iterate(ro, rd, gd, gc, t_rand=t_rand.to(DEV))
        finally:
            torch.rand = real
    np.testing.assert_allclose(es.params.detach().cpu().numpy(), g["grid_s1"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(ec.params.detach().cpu().numpy(), g["grid_c1"], rtol=1e-3, atol=2e-5)
    for k, v in dec.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["dec1__" + k.replace(".", "__")], rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("name", ["config1_256x64_torch_mlp_2x16", "config3_scannet_8192x96_zero_depth"])
def test_mapstep_baseline_configs_against_oracle(name):
    """
    BASELINE.json configs[0] and configs[2] at their full sizes, one mapping iteration against the CPU oracle on the same rays and
    random draws (configs[1] = test_mapstep_full_size_against_oracle):
      config 1: Replica room0, 256 rays x 64 samples (48 + 16), room0 tables log2T 16 / 19 at res 816, and the reference's torch-MLP
                decoders 32 -> 16 -> 16 -> out with biases (src/networks/decoders.py:74-84) -- the reference's own CPU-runnable case;
      config 3: ScanNet scene0000, 8192 rays x 96 samples (80 + 16), tables 16 / 16 at res 456, uncertainty-gated loss, 25 % of the
                rays without a depth measurement (the importance-sampling branch of src/utils/Renderer.py:104-130).
    Bars: z_vals 1e-4 (bit-exact for rays with a depth), rendered depth / colour 1e-3 relative (north star), loss 1e-3, table and
    decoder gradients 1e-3 norm-wise.
    """
    import unislam_amd as us
    torch.manual_seed(7)
    if name.startswith("config1"):
        bound, R, ns, ni, l2s, l2c, res, zero = BOUND, 256, 48, 16, 16, 19, 816, False
    else:
        bound, R, ns, ni, l2s, l2c, res, zero = O.load_bound([[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]]), 8192, 80, 16, 16, 16, 456, True
    S = ns + ni
    ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16,
                       "per_level_scale": O.per_level_scale(res)}
    dec = us.Decoders(_cfg(False, ns, ni), c_dim=32, hidden_size=16, truncation=0.06, n_blocks=2).to(DEV)
    es, ec = us.HashGridEncoding(3, ecfg(l2s)).to(DEV), us.HashGridEncoding(3, ecfg(l2c)).to(DEV)
    with torch.no_grad():
        es.params.copy_(torch.randn(es.params.shape) * 0.2); ec.params.copy_(torch.randn(ec.params.shape) * 0.2)
    g = torch.Generator().manual_seed(8)
    ro = bound.mean(1)[None].repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05
    rd = torch.randn(R, 3, generator=g); rd = rd / rd.norm(dim=-1, keepdim=True)
    far = O.bbox_far(ro, rd, bound)
    gd = torch.minimum(torch.rand(R, generator=g) * 3 + 0.5, 0.9 * far)          # every ray inside the box: the oracle call has no pre-filter
    if zero:
        gd[::4] = 0.0
    gc = torch.rand(R, 3, generator=g)
    n1, n0 = int((gd > 0).sum()), int((gd <= 0).sum())
    tr1, tr0, u0 = torch.rand(n1, S, generator=g), torch.rand(n0, ns, generator=g), torch.rand(n0, ni, generator=g)
    od = O.DecodersOracle(hidden_size=16, n_blocks=2); od.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    oes, oec = O.HashGridOracle(3, ecfg(l2s)), O.HashGridOracle(3, ecfg(l2c))
    with torch.no_grad():
        oes.params.copy_(es.params.cpu()); oec.params.copy_(ec.params.cpu())
    draws = {"z": tr1, "z_uni": tr0, "u": u0} if zero else {"z": tr1}
    ret_o = O.render_batch_ray(([oes], [oec]), od, rd, ro, 0.06, gd, bound, ns, ni, True, draws)
    loss_o = O.mapping_loss(ret_o, gd, gc, 0.06, W)
    loss_o.backward()
    step = us.MapStep(es, ec, dec, bound, ns, ni, 0.06, W, LR, max_rays=R)
    t_rand = torch.zeros(R, S); t_rand[gd > 0] = tr1
    loss = step.forward_backward(ro.to(DEV), rd.to(DEV), gd.to(DEV), gc.to(DEV), t_rand=t_rand.to(DEV), has_zero_depth=zero,
                                 zero_depth_draws=(tr0.to(DEV), u0.to(DEV)) if zero else None)
    term, unc, depth, rgb, sdf, z, dunc = [t.cpu() for t in step.rendered()]
    with_depth = gd > 0
    assert np.array_equal(z[with_depth].numpy(), ret_o[5][with_depth].numpy())              # depth-guided samples: bit-exact
    np.testing.assert_allclose(z.numpy(), ret_o[5].numpy(), rtol=1e-4, atol=1e-5)
    # zero-depth rays place their importance samples by inverting a cdf: a 1e-5 difference in an sdf moves a sample a little, and
    # the rendered values of THOSE rays follow; they are held norm-wise, the rays with a depth element-wise
    np.testing.assert_allclose(depth[with_depth].numpy(), ret_o[2].detach()[with_depth].numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(rgb[with_depth].numpy(), ret_o[3].detach()[with_depth].numpy(), rtol=1e-3, atol=1e-5)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(depth, ret_o[2].detach()) < 1e-3 and rel(rgb, ret_o[3].detach()) < 1e-3
    m_o = (gd > 0) & ((1 - ret_o[1].detach()) > 0.99)
    assert abs(int(step.stats[9]) - int(m_o.sum())) <= 2                                     # rays on the 0.99 opacity threshold may flip
    np.testing.assert_allclose(float(loss), float(loss_o), rtol=1e-3)
    for nm, g_hip, g_o in (("sdf", es.params.grad.cpu(), oes.params.grad), ("colour", ec.params.grad.cpu(), oec.params.grad)):
        assert rel(g_hip, g_o) < 1e-3, (nm, rel(g_hip, g_o))
    for (n, pa), (_, pb) in zip(od.named_parameters(), dec.named_parameters()):
        assert rel(pb.grad.cpu(), pa.grad) < 2e-3, (n, rel(pb.grad.cpu(), pa.grad))


def test_mapstep_config3_shape_scannet():
    """BASELINE configs[2]: ScanNet scene0000 tables (res 456, log2T 16/16), 8192 rays x 96 samples, uncertainty gating on, 25 % zero-depth rays"""
    import unislam_amd as us
    torch.manual_seed(0)
    bound = O.load_bound([[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]])
    assert O.get_resolution(bound, 0.02) == 456
    ecfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 16, "base_resolution": 16,
            "per_level_scale": O.per_level_scale(456)}
    es, ec = us.HashGridEncoding(3, ecfg).to(DEV), us.HashGridEncoding(3, ecfg).to(DEV)
    assert es.desc.n_params == 1697200
    with torch.no_grad():
        es.params.mul_(2000); ec.params.mul_(2000)
    dec = us.Decoders(_cfg(False, 80, 16), c_dim=32, truncation=0.06).to(DEV)
    R = 8192
    g = torch.Generator().manual_seed(3)
    ro = bound.mean(1)[None].repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05
    rd = torch.randn(R, 3, generator=g); rd = rd / rd.norm(dim=-1, keepdim=True)
    far = O.bbox_far(ro, rd, bound)
    gd = torch.minimum(torch.rand(R, generator=g) * 3 + 0.5, 0.9 * far); gd[::4] = 0.0
    gc = torch.rand(R, 3, generator=g)
    step = us.MapStep(es, ec, dec, bound, 80, 16, 0.06, W, dict(decoders=0.001, sdf_grid=0.02, color_grid=0.02), max_rays=R)
    losses = [float(step.iterate(ro.to(DEV), rd.to(DEV), gd.to(DEV), gc.to(DEV))) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    z = step.rendered()[5]
    assert bool((z[:, 1:] >= z[:, :-1]).all())                                # sorted samples incl. the importance-sampled zero-depth rays
    st = step.stats.cpu().numpy()
    assert st[8] == 3 * R and st[9] <= R * 0.75 + 1                           # colour over all rays, depth only where gt > 0 and opaque


@pytest.mark.parametrize("tcnn,hidden", [(False, 32), (True, 16)])
def test_mapstep_bf16_decoders_track_fp32(tcnn, hidden):
    """mlp_precision = bf16 (bf16 MFMA with split operands in the forward products, fp32 accumulation and parameters): same rays and
    draws as the fp32 step.  Rendered depth / colour stay within the north star's 1e-3 relative of the fp32 path -- element-wise --
    and the optimisation behaves the same."""
    import unislam_amd as us
    R, S = 1024, 64
    ro, rd, gd, gc = _rays(R, seed=5)
    t_rand = torch.rand(R, S, device=DEV)
    outs, losses = {}, {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(3)
        cfg = dict(_cfg(tcnn, 48, 16), model={"mlp_precision": prec})
        dec = us.Decoders(cfg, c_dim=32, hidden_size=hidden, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 48, 16, 0.06, W, LR, max_rays=R)
        assert step.desc_s.precision == (1 if prec == "bf16" else 0)
        l0 = float(step.forward_backward(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False))
        outs[prec] = [t.clone() for t in step.rendered()[:4]] + [es.params.grad.clone(), ec.params.grad.clone()]
        step.adam_step()
        ls = [l0] + [float(step.iterate(ro, rd, gd, gc, has_zero_depth=False)) for _ in range(25)]
        losses[prec] = ls
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    term_b, unc_b, depth_b, rgb_b, gs_b, gc_b = outs["bf16"]
    term_f, unc_f, depth_f, rgb_f, gs_f, gc_f = outs["fp32"]
    assert rel(depth_b, depth_f) < 1e-4 and rel(rgb_b, rgb_f) < 1e-4, (rel(depth_b, depth_f), rel(rgb_b, rgb_f))
    np.testing.assert_allclose(depth_b.cpu().numpy(), depth_f.cpu().numpy(), rtol=1e-3, atol=1e-5)       # the north-star bound, per ray
    np.testing.assert_allclose(rgb_b.cpu().numpy(), rgb_f.cpu().numpy(), rtol=1e-3, atol=1e-5)
    assert rel(gs_b, gs_f) < 2e-2 and rel(gc_b, gc_f) < 2e-2, (rel(gs_b, gs_f), rel(gc_b, gc_f))
    assert abs(losses["bf16"][0] - losses["fp32"][0]) < 2e-2 * abs(losses["fp32"][0])
    assert losses["bf16"][-1] < losses["bf16"][0] and abs(losses["bf16"][-1] - losses["fp32"][-1]) < 0.1 * abs(losses["fp32"][-1])


@pytest.mark.parametrize("tcnn", [False, True])
def test_mapstep_ray_gradients_match_autograd(tcnn):
    """backward(ray_grads=True): dL/d rays_o, dL/d rays_d of the mapping loss (what Mapper.py:358-374 joint_opt differentiates
    through cam_pose_to_matrix) against torch autograd through Renderer + Decoders + losses on the same rays and draws."""
    import unislam_amd as us
    dec, es, ec = _scene(us, tcnn)
    R, S = 257, 40
    ro, rd, gd, gc = _rays(R, outside=True)
    t_rand = torch.rand(R, S, device=DEV)
    rend = us.Renderer(_cfg(tcnn), types.SimpleNamespace(bound=BOUND, device=DEV, H=12, W=16, fx=10., fy=10., cx=7.5, cy=5.5))
    roa, rda = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
    inside = us.common.bbox_filter(ro, rd, gd, BOUND)
    ret = rend.render_batch_ray(([es], [ec]), dec, rda[inside], roa[inside], DEV, 0.06, gt_depth=gd[inside], t_rand=t_rand[inside])
    us.mapping_loss(ret, gd[inside], gc[inside], 0.06, W).backward()
    dec2, es2, ec2 = copy.deepcopy(dec), copy.deepcopy(es), copy.deepcopy(ec)
    step = us.MapStep(es2, ec2, dec2, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
    step.forward_backward(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False, ray_grads=True)
    g_o, g_d = step.ray_gradients()
    assert float(g_o[~inside].abs().max()) == 0.0 and float(g_d[~inside].abs().max()) == 0.0
    for a, b in ((g_o, roa.grad), (g_d, rda.grad)):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-4 * float(b.abs().max())), float((a - b).abs().max() / b.abs().max())


def test_mapstep_fixed_beta_tum_config():
    """configs/TUM_RGBD/tum.yaml:49 `learnable_beta: False`: beta is the python scalar 10 (decoders.py:86-89), no beta gradient."""
    import unislam_amd as us
    torch.manual_seed(0)
    dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06, learnable_beta=False).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(14)).to(DEV)
    with torch.no_grad():
        es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
    assert not isinstance(dec.beta, torch.nn.Parameter) and float(dec.beta) == 10
    dec2, es2, ec2 = copy.deepcopy(dec), copy.deepcopy(es), copy.deepcopy(ec)
    R, S = 200, 40
    ro, rd, gd, gc = _rays(R)
    t_rand = torch.rand(R, S, device=DEV)
    rend = us.Renderer(_cfg(False), types.SimpleNamespace(bound=BOUND, device=DEV, H=12, W=16, fx=10., fy=10., cx=7.5, cy=5.5))
    opt = torch.optim.Adam([{"params": list(dec2.parameters()), "lr": LR["decoders"]},
                            {"params": [es2.params], "lr": LR["sdf_grid"]}, {"params": [ec2.params], "lr": LR["color_grid"]}])
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
    for it in range(2):
        inside = us.common.bbox_filter(ro, rd, gd, BOUND)                   # Mapper.py:396-406
        ret = rend.render_batch_ray(([es2], [ec2]), dec2, rd[inside], ro[inside], DEV, 0.06, gt_depth=gd[inside], t_rand=t_rand[inside])
        loss_a = us.mapping_loss(ret, gd[inside], gc[inside], 0.06, W)
        opt.zero_grad(); loss_a.backward(); opt.step()
        loss_b = step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
        np.testing.assert_allclose(float(loss_b), float(loss_a.detach()), rtol=5e-5)
    assert torch.allclose(es.params, es2.params, rtol=1e-3, atol=1e-5) and torch.allclose(ec.params, ec2.params, rtol=1e-3, atol=1e-5)
    for (n, pa), (_, pb) in zip(dec2.named_parameters(), dec.named_parameters()):
        assert torch.allclose(pa, pb, rtol=1e-3, atol=1e-5), n


def test_mapstep_with_a_table_beyond_the_bin_budget():
    """a colour table of 2^22 entries per level does not fit the binned backward's 4096 bins: MapStep falls back to the sliced
    kernels by itself and still optimises"""
    import unislam_amd as us
    torch.manual_seed(0)
    dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(16)).to(DEV), us.HashGridEncoding(3, _ecfg(22)).to(DEV)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=1024)
    assert step.bwd_mode == 1 and step.ws is None
    ro, rd, gd, gc = _rays(1024, seed=3)
    losses = [float(step.iterate(ro, rd, gd, gc, has_zero_depth=False)) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_mapstep_graph_replay_equals_eager():
    """MapStep.capture / replay: the captured iteration (device-side Adam step count, in-place inputs) reproduces eager iterate()
    calls on the same rays and jitter, can be mixed with them, leaves the state alone while capturing, and is dropped by
    reset_optimizer."""
    import unislam_amd as us
    R, S = 512, 40
    ro, rd, gd, gc = _rays(R, seed=31, outside=True)
    t_rand = torch.rand(R, S, generator=torch.Generator().manual_seed(3)).to(DEV)
    outs = []
    for mode in ("eager", "graph", "mixed"):
        dec, es, ec = _scene(us, False, seed=30)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
        step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)               # one eager step first: moments are non-zero
        losses = []
        if mode != "eager":
            before = (step.flat.clone(), step.m.clone(), step.v.clone(), float(step.step_dev[0]))
            ins = step.capture(R, t_rand=True)
            assert torch.equal(step.flat, before[0]) and torch.equal(step.m, before[1]) and torch.equal(step.v, before[2])
            assert float(step.step_dev[0]) == before[3] == 1.0 and step.opt_step == 1
            for dst, src in zip(ins, (ro, rd, gd, gc, t_rand)):
                dst.copy_(src)
        for k in range(5):
            if mode == "eager" or (mode == "mixed" and k % 2 == 1):
                losses.append(float(step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)))
            else:
                losses.append(float(step.replay()))
        assert float(step.step_dev[0]) == 6.0 and step.opt_step == 6
        outs.append((step.flat.clone(), losses))
        if mode == "graph":
            # new rays written into the static inputs are what the next replay sees
            ro2, rd2, gd2, gc2 = _rays(R, seed=77)
            for dst, src in zip(ins, (ro2, rd2, gd2, gc2, t_rand)):
                dst.copy_(src)
            l_new = float(step.replay())
            dec_b, es_b, ec_b = _scene(us, False, seed=30)
            ref = us.MapStep(es_b, ec_b, dec_b, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
            for _ in range(6):
                ref.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
            np.testing.assert_allclose(l_new, float(ref.iterate(ro2, rd2, gd2, gc2, t_rand=t_rand, has_zero_depth=False)), rtol=1e-5)
            step.reset_optimizer()
            with pytest.raises(us.UniSlamHipError):
                step.replay()
    for flat, losses in outs[1:]:
        np.testing.assert_allclose(losses, outs[0][1], rtol=1e-6)
        assert torch.allclose(flat, outs[0][0], rtol=1e-6, atol=1e-8)
    # without t_rand the in-kernel generator draws differently in every replay (device-side counter)
    dec, es, ec = _scene(us, False, seed=30)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R)
    ins = step.capture(R)
    for dst, src in zip(ins, (ro, rd, gd, gc)):
        dst.copy_(src)
    step.replay(); z1 = step.rendered()[5].clone()
    step.replay(); z2 = step.rendered()[5].clone()
    assert not torch.equal(z1, z2) and bool((z1[:, 1:] >= z1[:, :-1]).all())


@pytest.mark.parametrize("pair", [(14, 15), (16, 19), (16, 16)])
def test_mapstep_joint_grids_equal_separate_grids(pair):
    """MapStep(joint=True) -- both encoders in one launch, both table gradients in one binned pass (csrc/hashgrid_joint.hip) -- against
    MapStep(joint=False) on the same rays and draws: bit-identical rendering, gradients equal up to one f64 -> f32 rounding, the same
    parameters after three Adam steps; rays the pre-filter drops and rays without depth included."""
    import unislam_amd as us
    R, S = 700, 40
    ro, rd, gd, gc = _rays(R, seed=21, zero_depth=True, outside=True)
    t_rand = torch.rand(R, S, device=DEV)
    res = {}
    for joint in (False, True):
        torch.manual_seed(4)
        dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(pair[0])).to(DEV), us.HashGridEncoding(3, _ecfg(pair[1])).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R, joint=joint)
        assert step.joint == joint
        n0 = int((gd <= 0).sum())
        draws = (torch.rand(n0, 32, generator=torch.Generator().manual_seed(1)).to(DEV), torch.rand(n0, 8, generator=torch.Generator().manual_seed(2)).to(DEV))
        loss = step.forward_backward(ro, rd, gd, gc, t_rand=t_rand, zero_depth_draws=draws)
        out = [t.clone() for t in step.rendered()] + [step.grad.clone(), loss.clone()]
        step.adam_step()
        for _ in range(2):
            step.forward_backward(ro, rd, gd, gc, t_rand=t_rand, zero_depth_draws=draws); step.adam_step()
        res[joint] = out + [step.flat.clone()]
    for k in range(7):
        assert torch.equal(res[True][k], res[False][k]), k                    # rendering: bit-identical
    ga, gb = res[True][7], res[False][7]
    assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-7 * float(gb.abs().max()))
    assert torch.equal(res[True][8], res[False][8])
    close = torch.isclose(res[True][9], res[False][9], rtol=1e-5, atol=1e-6)
    assert float((~close).float().mean()) < 1e-4


@pytest.mark.parametrize("joint", [True, False])
def test_mapstep_table_gradient_in_ranges(joint):
    """max_workspace_bytes: a batch whose table-gradient scratch would exceed the budget is walked in ranges of rays
    (us_hashgrid_bwd_joint_range / us_hashgrid_bwd_binned_range: first range OVERWRITE, the others add); same gradients and
    parameters as the one-pass step, with a workspace several times smaller."""
    import unislam_amd as us
    R, S = 1000, 40
    ro, rd, gd, gc = _rays(R, seed=33, outside=True)
    t_rand = torch.rand(R, S, device=DEV)
    res = {}
    for budget in (4 << 30, 48 << 20):
        torch.manual_seed(4)
        dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(16)).to(DEV), us.HashGridEncoding(3, _ecfg(19)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R, joint=joint, max_workspace_bytes=budget)
        if budget < (1 << 30):
            assert 0 < step.chunk_rays < R and step.ws_bytes <= budget, (step.chunk_rays, step.ws_bytes)
        else:
            assert step.chunk_rays == 0
        loss = step.forward_backward(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
        g = step.grad.clone()
        step.adam_step()
        step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
        res[budget] = (g, loss.clone(), step.flat.clone(), step.ws_bytes)
    a, b = res[4 << 30], res[48 << 20]
    assert b[3] < 0.6 * a[3]                                  # (one-grid kernels: 63 MB per scratch set -> two ranges of 31 MB)
    assert torch.equal(a[1], b[1])
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-6 * float(a[0].abs().max()))
    close = torch.isclose(a[2], b[2], rtol=1e-5, atol=1e-6)
    assert float((~close).float().mean()) < 1e-4


def test_bench_path_replay_equals_eager_and_oracle():
    """Exactly what bench.py builds for its headline -- MapStep joint, mlp_precision bf16, decoder pair, a MapWindow (joint_opt off) over
    16 keyframe pools whose captured graph holds ray assembly + sampling + the iteration -- replayed five times against five eager
    iterations on the same pixel indices and jitter (loss 1e-6, parameters allclose), and the first iteration against the CPU oracle
    (1e-3)."""
    import bench
    import unislam_amd as us
    bench.torch = torch
    bound = bench.load_bound(bench.ROOM0_BOUND)
    pls = bench.per_level_scale(int((bound[:, 1] - bound[:, 0]).max() / 0.01))
    R, S, nk = 4096, 64, bench.N_KEYFRAMES
    c2ws, pool_d, pool_c, pool_dirs = bench.keyframe_pools(nk, bound, 1000, DEV)
    P, n_per = pool_d.shape[1], R // nk
    g = torch.Generator().manual_seed(12)
    draws = [(torch.randint(P, (nk, n_per), generator=g).to(DEV), torch.rand(R, S, generator=g).to(DEV)) for _ in range(5)]

    def build():
        torch.manual_seed(0)
        cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
        dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                                "base_resolution": 16, "per_level_scale": pls}).to(DEV)
        es, ec = mk(16), mk(19)
        with torch.no_grad():                                   # a non-trivial surface instead of the initial U(+-1e-4)
            es.params.normal_(0.0, 0.1); ec.params.normal_(0.0, 0.1)
        step = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, bench.W, bench.LR, max_rays=R)
        return step, us.MapWindow(step, c2ws, pool_d, pool_c, pool_dirs, n_per, joint_opt=False, has_zero_depth=False), es, ec, dec

    # ---- eager
    step_e, win_e, es_e, ec_e, dec_e = build()
    assert step_e.joint and step_e._decoder_pair() and step_e.overlap
    # the first iteration also against the oracle: same rays, same jitter
    win_e.draw(draws[0][0])
    ro, rd, gd, gc = [t.cpu() for t in win_e.rays()]
    od = O.DecodersOracle(hidden_size=32, n_blocks=2); od.load_state_dict({k: v.cpu() for k, v in dec_e.state_dict().items()})
    oes, oec = O.HashGridOracle(3, es_e.encoding_config), O.HashGridOracle(3, ec_e.encoding_config)
    with torch.no_grad():
        oes.params.copy_(es_e.params.cpu()); oec.params.copy_(ec_e.params.cpu())
    assert bool((O.bbox_far(ro, rd, bound) >= gd).all())        # the pools' depths are clipped inside the box
    ret_o = O.render_batch_ray(([oes], [oec]), od, rd, ro, 0.06, gd, bound, 48, 16, True, {"z": draws[0][1].cpu()})
    loss_o = O.mapping_loss(ret_o, gd, gc, 0.06, bench.W)
    losses_e = []
    for idx, tr in draws:
        losses_e.append(float(win_e.iterate(idx, t_rand=tr)))
        if len(losses_e) == 1:
            depth, rgb = step_e.rendered()[2].cpu(), step_e.rendered()[3].cpu()
            np.testing.assert_allclose(depth.numpy(), ret_o[2].detach().numpy(), rtol=1e-3, atol=1e-5)
            np.testing.assert_allclose(rgb.numpy(), ret_o[3].detach().numpy(), rtol=1e-3, atol=1e-5)
            np.testing.assert_allclose(losses_e[0], float(loss_o.detach()), rtol=1e-3)
    # ---- the same five iterations replayed from the captured graph
    step_g, win_g, es_g, ec_g, dec_g = build()
    win_g.capture(t_rand=True, device_draw=False)
    losses_g = []
    for idx, tr in draws:
        win_g.t_rand.copy_(tr)
        losses_g.append(float(win_g.replay(idx)))
    np.testing.assert_allclose(losses_g, losses_e, rtol=1e-6)
    assert torch.allclose(step_g.flat, step_e.flat, rtol=1e-5, atol=1e-7), float((step_g.flat - step_e.flat).abs().max())
    assert float(step_g.step_dev[0]) == 5.0 == float(step_e.step_dev[0])
    # ... and with the draw inside the graph (what bench.py replays): fresh batches, finite losses
    step_d, win_d, *_ = build()
    win_d.capture()
    l_d = [float(win_d.replay()) for _ in range(3)]
    assert all(np.isfinite(l_d)) and float(step_d.step_dev[0]) == 3.0


def test_bf16_gradient_products_do_not_change_what_a_window_converges_to():
    """A 15-iteration mapping window (src/Mapper.py:366-445; Replica's `mapping.iters`) from identical state with fp32 and with bf16
    decoders (split-operand forward products, bf16 operands in the gradient products), same pixel draws and jitter: depth and colour
    of a held-out view rendered from the two resulting maps -- each with its own decoders -- agree within 1e-3 norm-wise."""
    import unislam_amd as us
    from unislam_amd.synthetic import SyntheticRoom
    R, iters, nf = 2048, 15, 4
    room = SyntheticRoom(n_frames=12, H=96, W=128, device=DEV)
    bound = O.load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]])
    frames = [room[k] for k in (0, 3, 6, 9)]
    held = room[5]
    c2ws = torch.stack([f[3] for f in frames])
    depths = torch.stack([f[2].reshape(-1) for f in frames]); colors = torch.stack([f[1].reshape(-1, 3) for f in frames])
    dirs = torch.stack([f[4].reshape(-1, 3) for f in frames])
    ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16,
                       "per_level_scale": O.per_level_scale(int((bound[:, 1] - bound[:, 0]).max() / 0.02))}
    outs = []
    for prec, joint in (("fp32", True), ("bf16", True), ("fp32", False), ("f16", True)):
        torch.manual_seed(0)
        dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": prec}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, ecfg(14)).to(DEV), us.HashGridEncoding(3, ecfg(15)).to(DEV)
        step = us.MapStep(es, ec, dec, bound, 32, 8, 0.06, W, LR, max_rays=R, joint=joint)
        win = us.MapWindow(step, c2ws, depths, colors, dirs, R // nf, joint_opt=False, has_zero_depth=False)
        g = torch.Generator().manual_seed(1)
        losses = []
        for _ in range(iters):
            idx = torch.randint(depths.shape[1], (nf, R // nf), generator=g).to(DEV)
            losses.append(float(win.iterate(idx, t_rand=torch.rand(R, 40, generator=g).to(DEV))))
        assert losses[-1] < 0.5 * losses[0]                                      # the window does converge
        rend = us.Renderer({"rendering": {"perturb": False, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid"},
                           types.SimpleNamespace(bound=bound, device=DEV, H=room.H, W=room.W, fx=room.fx, fy=room.fy, cx=room.cx, cy=room.cy))
        out = rend.render_img(([es], [ec]), dec, held[3], 0.06, DEV, gt_depth=held[2])
        outs.append((out[0].float(), out[1].float()))
    (d0, c0), (d1, c1), (d2, c2), (d3, c3) = outs
    dev = lambda a, b: float((a - b).norm() / a.norm())
    print("bf16 vs fp32:", dev(d0, d1), dev(c0, c1), " fp32 one-grid kernels vs fp32 joint kernels:", dev(d0, d2), dev(c0, c2),
          " f16 vs fp32:", dev(d0, d3), dev(c0, c3))
    assert dev(d0, d1) < 1e-3 and dev(c0, c1) < 1e-3
    assert dev(d0, d3) < 1e-3 and dev(c0, c3) < 1e-3                              # f16 operands, one product (US_PREC_F16)


def test_bf16_gradient_payload_does_not_change_what_a_window_converges_to():
    """The data-parallel step's payload options (dist.GradComm) on the convergence bar of the test above: a 15-iteration window from
    identical state, identical draws -- single process (fp32 gradients) against a process group whose colour-table segment ("bf16_colour")
    or whole gradient ("bf16") passes through bfloat16 on its way to Adam (1-rank RCCL group: every contribution is rounded once, as each
    rank's is).  Held-out view: depth / colour within 1e-3 norm-wise; with "bf16_colour" the depth does not move at all beyond the fp32
    noise floor (the geometry's gradients stay fp32)."""
    import socket
    import torch.distributed as dist
    import unislam_amd as us
    from unislam_amd.synthetic import SyntheticRoom
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        R, iters, nf = 2048, 15, 4
        room = SyntheticRoom(n_frames=12, H=96, W=128, device=DEV)
        bound = O.load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]])
        frames = [room[k] for k in (0, 3, 6, 9)]
        held = room[5]
        c2ws = torch.stack([f[3] for f in frames])
        depths = torch.stack([f[2].reshape(-1) for f in frames]); colors = torch.stack([f[1].reshape(-1, 3) for f in frames])
        dirs = torch.stack([f[4].reshape(-1, 3) for f in frames])
        ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16,
                           "per_level_scale": O.per_level_scale(int((bound[:, 1] - bound[:, 0]).max() / 0.02))}
        outs = []
        for group, comm in ((None, None), (True, "bf16_colour"), (True, "bf16"), (True, None)):
            torch.manual_seed(0)
            dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
            es, ec = us.HashGridEncoding(3, ecfg(14)).to(DEV), us.HashGridEncoding(3, ecfg(15)).to(DEV)
            step = us.MapStep(es, ec, dec, bound, 32, 8, 0.06, W, LR, max_rays=R, group=group, grad_comm=comm)
            step.rng_seed = 1234                                                 # (a group salts the sampler's seed with the rank; the draws here are given)
            win = us.MapWindow(step, c2ws, depths, colors, dirs, R // nf, joint_opt=False, has_zero_depth=False)
            g = torch.Generator().manual_seed(1)
            losses = []
            for _ in range(iters):
                idx = torch.randint(depths.shape[1], (nf, R // nf), generator=g).to(DEV)
                losses.append(float(win.iterate(idx, t_rand=torch.rand(R, 40, generator=g).to(DEV))))
            assert losses[-1] < 0.5 * losses[0]
            rend = us.Renderer({"rendering": {"perturb": False, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid"},
                               types.SimpleNamespace(bound=bound, device=DEV, H=room.H, W=room.W, fx=room.fx, fy=room.fy, cx=room.cx, cy=room.cy))
            out = rend.render_img(([es], [ec]), dec, held[3], 0.06, DEV, gt_depth=held[2])
            outs.append((out[0].float(), out[1].float()))
        dev = lambda a, b: float((a - b).norm() / a.norm())
        (d0, c0), (d1, c1), (d2, c2), (d3, c3) = outs
        print("payload bf16_colour vs fp32:", dev(d0, d1), dev(c0, c1), " bf16:", dev(d0, d2), dev(c0, c2), " fp32 over the group:", dev(d0, d3), dev(c0, c3))
        assert dev(d0, d3) < 1e-5 and dev(c0, c3) < 1e-5                          # the group by itself changes nothing but summation order
        assert dev(d0, d1) < 1e-5 and dev(c0, c1) < 1e-3                          # colour payload: the depth is untouched
        assert dev(d0, d2) < 1e-3 and dev(c0, c2) < 1e-3
    finally:
        dist.destroy_process_group()


def test_iterate_folds_the_decoder_reductions_into_their_adam_launch():
    """iterate() (single process, joint kernels, bf16 decoder pair): the decoder-gradient and beta reductions run inside the decoders'
    optimiser launch (us_mlp_reduce_pair_adam).  Same parameters, moments and gradients, bit for bit, as forward() + backward() +
    adam_step() with the separate reductions; five iterations."""
    import unislam_amd as us
    R, S = 700, 40
    ro, rd, gd, gc = _rays(R, seed=41, outside=True)
    g = torch.Generator().manual_seed(2)
    trs = [torch.rand(R, S, generator=g).to(DEV) for _ in range(5)]
    outs = []
    for folded in (True, "two launches", False):                # (True: the whole optimiser step in one launch, us_adam_step_model)
        torch.manual_seed(7)
        dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=R, deterministic=True)    # (no float atomics in hot bins)
        assert step.joint and step._decoder_pair() and step.overlap
        step.one_launch_adam = folded is True
        losses = []
        for tr in trs:
            if folded:
                losses.append(float(step.iterate(ro, rd, gd, gc, t_rand=tr, has_zero_depth=False)))
            else:
                step.forward(ro, rd, gd, gc, tr, False)
                losses.append(float(step.backward()))
                step.adam_step()
        torch.cuda.synchronize()
        outs.append((losses, step.flat.clone(), step.m.clone(), step.v.clone(), step.grad[:step.o_tab_s].clone()))
    a, a2, b = outs
    assert a[0] == b[0] and a2[0] == b[0]
    nd = step.o_tab_s
    for k in range(1, 4):
        assert torch.equal(a[k][:nd], b[k][:nd]) and torch.equal(a2[k][:nd], b[k][:nd]), k      # decoders + beta: parameters and moments bit for bit
        # (the tables do not take part in the change; their f64 sums are order-free up to rare last-bit flips of entries of 1e-20)
        assert torch.allclose(a[k][nd:], b[k][nd:], rtol=1e-6, atol=1e-12) and torch.allclose(a2[k][nd:], b[k][nd:], rtol=1e-6, atol=1e-12), k
    assert torch.equal(a[4], a2[4])
    # the decoder gradients of the last iteration are in the gradient buffer either way (the unfolded optimiser pass clears them: compare
    # against a recomputation)
    assert float(a[4].abs().max()) > 0

