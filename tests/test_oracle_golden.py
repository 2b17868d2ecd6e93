"""
CPU suite: the oracle (oracle/unislam_oracle.py) against golden vectors captured from the reference's own
Python by oracle/gen_golden.py.  This is what pins the oracle; the GPU parity tests
then compare the HIP path with the oracle.
"""
import numpy as np
import pytest
import torch

import unislam_oracle as O

T = torch.from_numpy
BOUND = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_bound_and_resolution():
    # room0 bound enlargement and desired resolution
    assert np.allclose(BOUND.numpy(), [[-1.0, 7.16], [-1.3, 3.74], [-1.7, 1.42]], atol=1e-5)
    assert O.get_resolution(BOUND, 0.01) == 816
    assert abs(O.per_level_scale(816) - 1.2996847159) < 1e-9
    b2 = O.load_bound([[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]])   # ScanNet scene0000
    assert O.get_resolution(b2, 0.02) == 456


def test_g1_rays(golden):
    g = golden("g1_rays")
    H, W, fx, fy, cx, cy = g["intr"]; H, W = int(H), int(W)
    close(O.get_camera_rays(H, W, fx, fy, cx, cy), g["cam"])
    ro, rd = O.get_rays(H, W, fx, fy, cx, cy, T(g["c2w"])[0])
    close(ro, g["rays_o"]); close(rd, g["rays_d"])
    s1 = O.get_samples(2, H - 2, 3, W - 3, 7, fx, fy, cx, cy, T(g["c2w"])[:1], T(g["depths"])[:1], T(g["colors"])[:1], T(g["s1_idx"]))
    for a, k in zip(s1, ("s1_o", "s1_d", "s1_depth", "s1_color")):
        close(a, g[k])
    s3 = O.get_samples(0, H, 0, W, 5, fx, fy, cx, cy, T(g["c2w"]), T(g["depths"]), T(g["colors"]), T(g["s3_idx"]))
    for a, k in zip(s3, ("s3_o", "s3_d", "s3_depth", "s3_color")):
        close(a, g[k])
    sa = O.get_samples_all(6, T(g["c2w"]), T(g["pool_d"]), T(g["pool_c"]), T(g["pool_r"]), T(g["sa_idx"]))
    for a, k in zip(sa, ("sa_o", "sa_d", "sa_depth", "sa_color")):
        close(a, g[k])


@pytest.mark.parametrize("ns,ni", [(32, 8), (48, 8)])
def test_g2_zsample(golden, ns, ni):
    g = golden("g2_zsample")
    gt = T(g["gt_depth"])[:, None]
    z0 = O.sample_z_with_depth(gt, float(g["truncation"]), ns, ni, False)
    assert np.array_equal(z0.numpy(), g[f"z_{ns}_{ni}_0"])          # bit-exact: same torch ops
    z1 = O.sample_z_with_depth(gt, float(g["truncation"]), ns, ni, True, T(g[f"trand_{ns}_{ni}"]))
    assert np.array_equal(z1.numpy(), g[f"z_{ns}_{ni}_1"])


@pytest.mark.parametrize("tag", ["b10", "b73"])
def test_g3_composite(golden, tag):
    g = golden("g3_composite")
    raw = T(g[f"{tag}_raw"]).requires_grad_(True)
    beta = T(g[f"{tag}_beta"]).requires_grad_(True)
    z = O.sample_z_with_depth(T(g[f"{tag}_gt"])[:, None], float(g["truncation"]), 32, 8, False)
    assert np.array_equal(z.numpy(), g[f"{tag}_z"])
    term, unc, depth, rgb, sdf, z2, dunc = O.composite(raw, z, beta)
    outs = dict(term=term, unc=unc, depth=depth, rgb=rgb, dunc=dunc)
    for k, v in outs.items():
        close(v, g[f"{tag}_{k}"], rtol=1e-5, atol=1e-6)
    sum((T(g[f"{tag}_probe_{k}"]) * v).sum() for k, v in outs.items()).backward()
    close(raw.grad, g[f"{tag}_draw"], rtol=1e-4, atol=1e-5)
    close(beta.grad, g[f"{tag}_dbeta"], rtol=1e-4, atol=1e-5)


def _load_dec(g, prefix, tcnn=False):
    dec = O.DecodersOracle(tcnn_network=tcnn)
    sd = {k[len(prefix):].replace("__", "."): T(v) for k, v in g.items() if k.startswith(prefix)}
    dec.load_state_dict(sd)
    return dec


def test_g4_decoders(golden):
    g = golden("g4_decoders")
    dec = O.DecodersOracle()
    sd = {k.replace("__", "."): T(g[k]) for k in g if (k.startswith(("linears", "c_linears", "output_linear", "c_output_linear")) or k == "beta")}
    dec.load_state_dict(sd)
    fs = T(g["feat_s"]).requires_grad_(True); fc = T(g["feat_c"]).requires_grad_(True)
    sr = ([lambda p: fs], [lambda p: fc])
    p = torch.rand(fs.shape[0], 3)
    sdf = dec.get_raw_sdf(p, sr); rgb = dec.get_raw_rgb(p, sr)
    close(sdf, g["sdf"]); close(rgb, g["rgb"])
    ((sdf * T(g["probe_s"])).sum() + (rgb * T(g["probe_c"])).sum()).backward()
    close(fs.grad, g["dfeat_s"], 1e-4, 1e-6); close(fc.grad, g["dfeat_c"], 1e-4, 1e-6)
    for n, p_ in dec.named_parameters():
        k = "grad__" + n.replace(".", "__")
        if k in g:
            close(p_.grad, g[k], 1e-4, 1e-5)


def test_g5_losses(golden):
    g = golden("g5_losses")
    sdf = T(g["sdf"]).requires_grad_(True); z = T(g["z"]); gt = T(g["gt"]); tr = float(g["truncation"])
    l = O.sdf_losses(sdf, z, gt, tr, 5, 200, 10)
    close(l, g["loss_map"]); l.backward(); close(sdf.grad, g["dsdf"], 1e-4, 1e-7)
    close(O.sdf_losses(sdf.detach(), z, gt, tr, 10, 200, 50), g["loss_trk"])
    ln = O.sdf_losses(sdf.detach(), T(g["z_far"]), gt, tr, 5, 200, 10)
    assert np.isnan(g["loss_nan"]) and torch.isnan(ln)              # torch.mean of an empty selection


def test_g6_sample_pdf_and_zero_depth(golden):
    g = golden("g6_zerodepth")
    s = O.sample_pdf(T(g["pdf_mid"]), T(g["pdf_w"]), g["pdf_u"].shape[1], T(g["pdf_u"]))
    close(s, g["pdf_samples"])
    dec = _load_dec(g, "dec__")
    es, ec = _grid(g["grid_s"]), _grid(g["grid_c"])
    torch.manual_seed(int(g["seed"]))
    ret = O.render_batch_ray(([es], [ec]), dec, T(g["rays_d"]), T(g["rays_o"]), 0.06, T(g["gt_depth"]), BOUND, 32, 8, True)
    close(ret[5], g["z_vals"], 1e-5, 1e-6); close(ret[2], g["depth"], 1e-4, 1e-5); close(ret[3], g["rgb"], 1e-4, 1e-5)


def _grid(params, log2T=10, res=64):
    enc = O.HashGridOracle(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2T,
                               "base_resolution": 16, "per_level_scale": O.per_level_scale(res)})
    with torch.no_grad():
        enc.params.copy_(T(params))
    return enc


def test_g7_render_fwd_bwd(golden):
    g = golden("g7_render")
    dec = _load_dec(g, "dec__")
    es, ec = _grid(g["grid_s"]), _grid(g["grid_c"])
    ro = T(g["rays_o"]).requires_grad_(True); rd = T(g["rays_d"]).requires_grad_(True)
    torch.manual_seed(int(g["seed"]))
    term, unc, depth, rgb, sdf, z, dunc = O.render_batch_ray(([es], [ec]), dec, rd, ro, 0.06, T(g["gt_depth"]), BOUND, 32, 8, True)
    outs = dict(term=term, unc=unc, depth=depth, rgb=rgb, sdf=sdf, dunc=dunc)
    assert np.array_equal(z.numpy(), g["z_vals"])
    for k, v in outs.items():
        close(v, g[k], 1e-5, 1e-6)
    sum((T(g["probe_" + k]) * v).sum() for k, v in outs.items()).backward()
    close(es.params.grad, g["g_grid_s"], 1e-4, 1e-6); close(ec.params.grad, g["g_grid_c"], 1e-4, 1e-6)
    close(ro.grad, g["g_rays_o"], 1e-3, 1e-4); close(rd.grad, g["g_rays_d"], 1e-3, 1e-4)
    for n, p_ in dec.named_parameters():
        close(p_.grad, g["gdec__" + n.replace(".", "__")], 1e-4, 1e-5)


def _tracking_iter(g, mode):
    H, W, fx, fy, cx, cy = g["intr"]; H, W = int(H), int(W)
    eh, ew = int(g["edge"][0]), int(g["edge"][1])
    dec = _load_dec(g, "dec__")
    for p in dec.parameters():
        p.requires_grad_(False)
    es, ec = _grid(g["grid_s"]), _grid(g["grid_c"])
    pose = T(g["pose"]).clone().requires_grad_(True)
    torch.manual_seed(int(g["seed"]))
    c2w = O.cam_pose_to_matrix(pose)
    ro, rd, gd, gc = O.get_samples(eh, H - eh, ew, W - ew, int(g["n"]), fx, fy, cx, cy, c2w, T(g["gt_depth"]), T(g["gt_color"]))
    with torch.no_grad():
        inside = (O.bbox_far(ro, rd, BOUND) >= gd) & (gd > 0)       # Tracker.py:177-184
    ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
    ret = O.render_batch_ray(([es], [ec]), dec, rd, ro, 0.06, gd, BOUND, 32, 8, True)
    loss = O.tracking_loss(ret, gd, gc, 0.06, dict(fs=10, center=200, tail=50, color=5, depth=1), mode)
    loss.backward()
    return loss, ret[1], pose.grad, es.params.grad


@pytest.mark.parametrize("mode", ["original", "no_mask"])
def test_g8_tracking_iteration(golden, mode):
    g = golden("g8_tracking")
    loss, unc, gpose, ggrid = _tracking_iter(g, mode)
    close(loss, g[f"{mode}_loss"], 1e-5, 1e-6); close(unc, g[f"{mode}_unc"], 1e-5, 1e-7)
    close(gpose, g[f"{mode}_gpose"], 1e-3, 1e-4); close(ggrid, g[f"{mode}_ggrid_s"], 1e-4, 1e-6)


def test_g9_mapping_two_iterations(golden):
    g = golden("g9_mapping")
    H, W, fx, fy, cx, cy = g["intr"]; H, W = int(H), int(W)
    dec = _load_dec(g, "dec0__")
    es, ec = _grid(g["grid_s0"]), _grid(g["grid_c0"])
    gt_depth, gt_color, c2w = T(g["gt_depth"]), T(g["gt_color"]), T(g["c2w"])
    cam = O.get_camera_rays(H, W, fx, fy, cx, cy)
    torch.manual_seed(int(g["seed"]))
    idx = torch.randperm(H * W)[:int(H * W * 0.1)]                 # Mapper.py:333-335 (10% pixel pool)
    pool_c, pool_d, pool_r = gt_color.reshape(-1, 3)[idx][None], gt_depth.reshape(-1)[idx][None], cam.reshape(-1, 3)[idx][None]
    f = float(g["lr_factor"])
    opt = torch.optim.Adam([{"params": list(dec.parameters()), "lr": 0.001 * f},       # Mapper.py:111-139
                            {"params": [es.params], "lr": 0.05 * f}, {"params": [ec.params], "lr": 0.05 * f}])
    w = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
    for _ in range(int(g["iters"])):
        ro, rd, gd, gc = O.get_samples_all(int(g["pixels"]), c2w[None], pool_d, pool_c, pool_r)
        O.mapping_iteration(([es], [ec]), dec, opt, ro, rd, gd, gc, BOUND, 0.06, 32, 8, w, "original", True)
    close(es.params, g["grid_s1"], 1e-4, 1e-6); close(ec.params, g["grid_c1"], 1e-4, 1e-6)
    for n, v in dec.state_dict().items():
        close(v, g["dec1__" + n.replace(".", "__")], 1e-4, 1e-6)


def test_g10_keyframe_selection(golden):
    """unislam_amd.slam.keyframe_selection_LC (host logic, torch ops only) against Mapper.keyframe_selection_LC of the reference"""
    from unislam_amd.slam import keyframe_selection_LC
    g = golden("g10_keyframes")
    H, W, fx, fy, cx, cy = [float(v) for v in g["intr"]]
    cam = (int(H), int(W), fx, fy, cx, cy)
    est, kfs = T(g["est"]), [int(k) for k in g["keyframe_list"]]
    for tag in ("plain", "loop", "back"):
        kl = kfs[:int(g[f"{tag}_n_kf"])]
        torch.manual_seed(int(g[f"{tag}_seed"]))
        sel, pct, loop = keyframe_selection_LC(len(kl) - 2, int(g[f"{tag}_idx"]), T(g["gt_color"]), T(g["gt_depth"]), T(g["c2w"]), 5, kl, est,
                                               cam, "cpu", tracking_back=bool(g[f"{tag}_tb"]), activated_mapping_mode=True, LC=True)
        assert [int(v) for v in sel] == [int(v) for v in g[f"{tag}_sel"]], tag
        assert int(loop) == int(g[f"{tag}_lc"])


def g14_window(g, tag):
    """the window of fixture g14 (pools, poses, the reference's random stream positioned after the current frame's randperm):
    (c2ws [b,4,4], depths [b,P], colors [b,P,3], dirs [b,P,3], n_per, extra, frames) -- shared with the GPU test"""
    H, W, fx, fy, cx, cy = g["intr"]; H, W = int(H), int(W)
    frames = [int(f) for f in g[f"{tag}_frames"]]
    kc, kd_, kcol, kdir = T(g[f"{tag}_kf_c2w"]), T(g[f"{tag}_kf_depth"]), T(g[f"{tag}_kf_color"]), T(g[f"{tag}_kf_dirs"])
    cam = O.get_camera_rays(H, W, fx, fy, cx, cy)
    torch.manual_seed(int(g[f"{tag}_seed"]))
    ind = torch.randperm(H * W)[:int(H * W * 0.1)]                 # Mapper.py:333-335: the current frame's pool
    cur_d, cur_c = T(g[f"{tag}_cur_depth"]).reshape(-1)[ind], T(g[f"{tag}_cur_color"]).reshape(-1, 3)[ind]
    depths = torch.cat([kd_[frames], cur_d[None]]); colors = torch.cat([kcol[frames], cur_c[None]])
    dirs = torch.cat([kdir[frames], cam.reshape(-1, 3)[ind][None]]); c2ws = torch.cat([kc[frames], T(g[f"{tag}_cur_c2w"])[None]])
    b = len(frames) + 1
    n_per = int(g[f"{tag}_pixels"]) // b                           # Mapper.py:315
    extra = (10, 200) if int(g[f"{tag}_n_kf"]) > 20 else None     # Mapper.py:385-393
    return c2ws, depths, colors, dirs, n_per, extra, frames


@pytest.mark.parametrize("tag", ["w6", "w12x"])
def test_g14_mapping_with_joint_pose_optimisation(golden, tag):
    """oracle restatement of Mapper.optimize_mapping with joint_opt (poses as a fourth Adam group) against the reference's run"""
    g = golden("g14_mapping_joint")
    w = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
    for iters in (1, 2):
        dec = _load_dec(g, "dec0__")
        es, ec = _grid(g["grid_s0"]), _grid(g["grid_c0"])
        c2ws, depths, colors, dirs, n_per, extra, frames = g14_window(g, tag)      # (re-seeds the random stream)
        b, P = depths.shape
        poses = torch.nn.Parameter(O.matrix_to_cam_pose(c2ws[1:]))
        opt = torch.optim.Adam([{"params": list(dec.parameters()), "lr": 0.001}, {"params": [es.params], "lr": 0.05},
                                {"params": [ec.params], "lr": 0.05}, {"params": [poses], "lr": float(g["cam_lr"])}])
        for _ in range(iters):
            idx = torch.randint(P, (n_per * b,)).reshape(b, -1)
            idx2 = torch.randint(P, (extra[1] * extra[0],)).reshape(extra[0], -1) if extra else None
            ro, rd, gd, gc = O.window_rays(c2ws[0], poses, depths, colors, dirs, idx, extra, idx2)
            O.mapping_iteration(([es], [ec]), dec, opt, ro, rd, gd, gc, BOUND, 0.06, 32, 8, w, "original", True)
        pre = f"{tag}_i{iters}_"
        close(poses, g[pre + "poses"], 1e-5, 1e-6)
        if iters == 1:
            gp = g[pre + "g_poses"]
            close(poses.grad, gp, 1e-3, 1e-4 * float(np.abs(gp).max())); close(dec.beta.grad, g[pre + "g_beta"], 1e-3, 1e-5)
            if tag == "w6":
                close(es.params.grad, g[pre + "g_grid_s"], 1e-4, 1e-6); close(ec.params.grad, g[pre + "g_grid_c"], 1e-4, 1e-6)
        else:
            close(es.params, g[pre + "grid_s"], 1e-4, 1e-6); close(ec.params, g[pre + "grid_c"], 1e-4, 1e-6)
            for n, v in dec.state_dict().items():
                close(v, g[pre + "dec__" + n.replace(".", "__")], 1e-4, 1e-6)
            c2 = torch.cat([c2ws[0:1], O.cam_pose_to_matrix(poses.detach())])
            close(c2[-1], g[pre + "cur_c2w"], 1e-5, 1e-6)
            close(c2[:-1], g[pre + "kf_c2w"][frames], 1e-5, 1e-6)
