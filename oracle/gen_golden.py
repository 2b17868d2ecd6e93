"""
oracle/gen_golden.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container (needs /root/reference).

Imports the reference's own Python (third-party modules that are absent here are stubbed in
sys.modules, in memory), runs its functions on CPU with fixed seeds and stores inputs + outputs
as small .npz fixtures under tests/golden/.  The reference source never travels; the fixtures are
data.  Re-run:  python oracle/gen_golden.py      (ALL generators, in file order: g4 / g7 / g8 / g9 take their decoders' initial weights
from torch's global generator as the generators before them left it; every other fixture, g14 included, is the same when its generator
runs alone.  tests/test_golden_regen.py regenerates everything into a temporary folder and compares bit for bit.)

Fixtures:
  g1_rays      get_camera_rays / get_rays / get_rays_from_uv / get_samples / get_samples_all
  g2_zsample   Renderer z-sampling (depth-guided) with and without perturbation
  g3_composite sdf2alpha + compositing 7-tuple + gradients
  g4_decoders  Decoders torch-MLP path: outputs + parameter / input gradients
  g5_losses    sdf_losses (incl. empty-mask NaN) ; mapping/tracking loss come through g8/g9
  g6_zerodepth zero-depth branch + sample_pdf (un-normalised pdf quirk)
  g7_render    render_batch_ray fwd+bwd through reference Renderer+Decoders with the oracle's CPU
               hash grid plugged in as scene_rep (pins everything AROUND the encoder)
  g8_tracking  Tracker.optimize_tracking, one full iteration (dummy self)
  g9_mapping   Mapper.optimize_mapping, two full iterations incl. Adam (dummy self)
  g10_keyframes Mapper.keyframe_selection_LC (dummy self): overlap ranking, loop-closure window, tracking-back draw
  g11_datasets  src/utils/datasets.py: the pose parsers of Replica / ScanNet / Azure / RGBDataset and TUM_RGBD.loadtum (time-stamp
                association, frame-rate thinning, first-frame-relative poses) on small text files written here; the image decode of
                __getitem__ needs OpenCV, which the image lacks, and is not captured
  g12_ate       src/tools/eval_ate.py: align (Horn), associate, evaluate_ate, pose_evaluation on generated trajectories
  g14_mapping_joint  Mapper.optimize_mapping with joint_opt = True (dummy self): a 6-frame window, and a 12-frame window of a 22-keyframe
                list with the 10 x 200 extra rays of Mapper.py:385-393; one iteration (pose / table gradients, state after Adam) and two
                iterations (final tables, decoders, poses); pytorch3d's two quaternion helpers are the oracle's restatement
  g15_sequence  the LOOP: Tracker.run / Mapper.run bodies alternating over 34 frames of the analytic room (every frame tracked, mapped and kept
                as a keyframe: joint_opt from the fifth keyframe, the extra rays beyond 20), optimize_tracking / optimize_mapping /
                keyframe_selection_LC being the reference's own methods: estimated trajectory, keyframe list, ATE (BASELINE configs[4]'s "vs reference")
  g13_scene     src/UNISLAM.py update_cam / load_bound / get_resolution (dummy self) and the per_level_scale line of get_encoder, and
                src/config.py load_config, for the room0 / scene0000 / fr1_desk settings (their numbers are inputs of the fixture)
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.environ.get("US_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")     # (tests/test_golden_regen.py writes to a temp dir)
REF = "/root/reference"

sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import unislam_oracle as O  # noqa: E402

for m in ["pytorch3d", "pytorch3d.transforms", "tinycudann", "colorama", "cv2", "skimage", "skimage.metrics",
          "open3d", "trimesh", "torchmetrics", "pytorch_msssim", "torchmetrics.image", "torchmetrics.image.lpip"]:
    sys.modules[m] = MagicMock()
# pytorch3d is absent: the pose helper is build-owned (parity unpinned for it); give the reference our restatement
sys.modules["pytorch3d.transforms"].quaternion_to_matrix = O.quaternion_to_matrix
sys.modules["pytorch3d.transforms"].matrix_to_quaternion = O.matrix_to_quaternion
sys.path.insert(0, REF)
import warnings  # noqa: E402
warnings.simplefilter("ignore")
from src import common as RC  # noqa: E402
from src.utils.Renderer import Renderer as RefRenderer  # noqa: E402
from src.networks.decoders import Decoders as RefDecoders  # noqa: E402
from src.Mapper import Mapper as RefMapper  # noqa: E402
from src.Tracker import Tracker as RefTracker  # noqa: E402
import src.Tracker as RT  # noqa: E402
import src.Mapper as RM  # noqa: E402

RC.quaternion_to_matrix = O.quaternion_to_matrix
RC.matrix_to_quaternion = O.matrix_to_quaternion
DEV = "cpu"


def npz(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in kw.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(f"{name}: {sum(a.nbytes for a in arrs.values())/1024:.1f} KiB raw, keys={list(arrs)}")


def rand_c2w(g, n):
    q = torch.randn(n, 4, generator=g); q = q / q.norm(dim=-1, keepdim=True)
    c = torch.eye(4).repeat(n, 1, 1)
    c[:, :3, :3] = O.quaternion_to_matrix(q); c[:, :3, 3] = torch.randn(n, 3, generator=g) * 0.3
    return c


BOUND = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])      # Replica room0 (configs/Replica/room0.yaml:3)


def make_cfg(n_strat, n_imp, perturb, tcnn_network=False):
    return {"rendering": {"perturb": perturb, "n_stratified": n_strat, "n_importance": n_imp},
            "scale": 1, "grid_mode": "hash_grid", "grid": {"tcnn_network": tcnn_network}}


def make_renderer(cfg, H=12, W=16, fx=10., fy=10., cx=7.5, cy=5.5):
    u = types.SimpleNamespace(bound=BOUND, device=DEV, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy)
    return RefRenderer(cfg, u)


def small_grid(seed, log2T=10, res=64, amp=0.5):
    enc = O.HashGridOracle(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
                               "log2_hashmap_size": log2T, "base_resolution": 16,
                               "per_level_scale": O.per_level_scale(res)})
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        enc.params.copy_((torch.rand(enc.params.shape, generator=g) * 2 - 1) * amp)
    return enc


def g1():
    g = torch.Generator().manual_seed(1)
    H, W, fx, fy, cx, cy = 12, 16, 11.0, 12.0, 7.5, 5.5
    cam = RC.get_camera_rays(H, W, fx, fy, cx, cy)
    c2w = rand_c2w(g, 3)
    ro, rd = RC.get_rays(H, W, fx, fy, cx, cy, c2w[0], DEV)
    depths = torch.rand(3, H, W, generator=g) * 3; colors = torch.rand(3, H, W, 3, generator=g)
    # get_samples (tracking): crop 2..H-2, 3..W-3; b=1 is what the tracker uses, b=3 also exercised
    torch.manual_seed(11)
    s1 = RC.get_samples(2, H - 2, 3, W - 3, 7, H, W, fx, fy, cx, cy, c2w[:1], depths[:1], colors[:1], DEV)
    torch.manual_seed(11)
    idx1 = torch.randint((H - 4) * (W - 6), (7 * 1,))
    torch.manual_seed(12)
    s3 = RC.get_samples(0, H, 0, W, 5, H, W, fx, fy, cx, cy, c2w, depths, colors, DEV)
    torch.manual_seed(12)
    idx3 = torch.randint(H * W, (5 * 3,))
    # get_samples_all (mapping): pools of P pixels
    P = 40
    pool_idx = torch.stack([torch.randperm(H * W, generator=g)[:P] for _ in range(3)])
    pd = torch.gather(depths.reshape(3, -1), 1, pool_idx)
    pc = torch.gather(colors.reshape(3, -1, 3), 1, pool_idx.unsqueeze(-1).expand(-1, -1, 3))
    pr = cam.reshape(-1, 3)[pool_idx]
    torch.manual_seed(13)
    sa = RC.get_samples_all(0, H, 0, W, 6, H, W, fx, fy, cx, cy, c2w, pd, pc, DEV, pr)
    torch.manual_seed(13)
    idxa = torch.randint(P, (6 * 3,)).reshape(3, -1)
    npz("g1_rays", intr=np.array([H, W, fx, fy, cx, cy]), cam=cam, c2w=c2w, rays_o=ro, rays_d=rd, depths=depths,
        colors=colors, s1_idx=idx1, s1_o=s1[0], s1_d=s1[1], s1_depth=s1[2], s1_color=s1[3],
        s3_idx=idx3, s3_o=s3[0], s3_d=s3[1], s3_depth=s3[2], s3_color=s3[3],
        pool_d=pd, pool_c=pc, pool_r=pr, sa_idx=idxa, sa_o=sa[0], sa_d=sa[1], sa_depth=sa[2], sa_color=sa[3])


class _ConstDecoders:
    """stand-in decoders so that render_batch_ray returns z_vals without any model in the loop"""
    beta = 10

    def __call__(self, pts, scene_rep):
        return torch.zeros(*pts.shape[:-1], 4)


def g2():
    g = torch.Generator().manual_seed(2)
    R = 32
    gt = torch.rand(R, generator=g) * 3 + 0.3
    ro = torch.zeros(R, 3); rd = torch.randn(R, 3, generator=g)
    out = {"gt_depth": gt, "truncation": np.float64(0.06)}
    for (ns, ni) in [(32, 8), (48, 8)]:
        for perturb in (False, True):
            r = make_renderer(make_cfg(ns, ni, perturb))
            torch.manual_seed(21)
            z = r.render_batch_ray(None, _ConstDecoders(), rd, ro, DEV, 0.06, gt_depth=gt)[5]
            key = f"z_{ns}_{ni}_{int(perturb)}"
            out[key] = z
            if perturb:
                torch.manual_seed(21)
                out[f"trand_{ns}_{ni}"] = torch.rand(R, ns + ni)
    npz("g2_zsample", **out)


class _FixedRawDecoders:
    """decoders stub that returns a fixed raw[R,S,4] leaf: drives the reference's compositing lines in isolation"""

    def __init__(self, raw, beta):
        self.raw, self.beta = raw, beta

    def __call__(self, pts, scene_rep):
        return self.raw


def g3():
    g = torch.Generator().manual_seed(3)
    R, S = 24, 40
    out = {"truncation": np.float64(0.06)}
    for tag, beta0 in [("b10", 10.0), ("b73", 7.3)]:
        raw = torch.randn(R, S, 4, generator=g) * 0.5
        raw[..., :3] = torch.sigmoid(raw[..., :3])
        raw[..., 3] = torch.tanh(raw[..., 3] + torch.linspace(1.5, -1.5, S))       # sdf crossing zero along the ray
        raw.requires_grad_(True)
        beta = torch.nn.Parameter(torch.tensor([beta0]))
        gt = torch.rand(R, generator=g) * 3 + 0.3
        ro = torch.zeros(R, 3); rd = torch.randn(R, 3, generator=g)
        r = make_renderer(make_cfg(32, 8, False))
        term, unc, depth, rgb, sdf, z, dunc = r.render_batch_ray(None, _FixedRawDecoders(raw, beta), rd, ro, DEV, 0.06, gt_depth=gt)
        outs = dict(term=term, unc=unc, depth=depth, rgb=rgb, dunc=dunc)
        probes = {k: torch.randn(v.shape, generator=g) for k, v in outs.items()}
        sum((probes[k] * v).sum() for k, v in outs.items()).backward()
        out.update({f"{tag}_raw": raw, f"{tag}_gt": gt, f"{tag}_beta": beta, f"{tag}_z": z, f"{tag}_draw": raw.grad,
                    f"{tag}_dbeta": beta.grad, f"{tag}_sdf": sdf})
        out.update({f"{tag}_{k}": v for k, v in outs.items()})
        out.update({f"{tag}_probe_{k}": v for k, v in probes.items()})
    npz("g3_composite", **out)


def g4():
    g = torch.Generator().manual_seed(4)
    cfg = make_cfg(32, 8, False)
    dec = RefDecoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
    N = 96
    feat_s = (torch.randn(N, 32, generator=g) * 0.7).requires_grad_(True)
    feat_c = (torch.randn(N, 32, generator=g) * 0.7).requires_grad_(True)
    sr = ([lambda p: feat_s], [lambda p: feat_c])
    p = torch.rand(N, 3, generator=g)
    sdf = dec.get_raw_sdf(p, sr); rgb = dec.get_raw_rgb(p, sr)
    ps, pc = torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    ((sdf * ps).sum() + (rgb * pc).sum()).backward()
    sd = {k.replace(".", "__"): v for k, v in dec.state_dict().items()}
    gd = {"grad__" + k.replace(".", "__"): v.grad for k, v in dec.named_parameters() if v.grad is not None}
    npz("g4_decoders", feat_s=feat_s, feat_c=feat_c, sdf=sdf, rgb=rgb, probe_s=ps, probe_c=pc,
        dfeat_s=feat_s.grad, dfeat_c=feat_c.grad, **sd, **gd)


def g5():
    g = torch.Generator().manual_seed(5)
    R, S, tr = 20, 40, 0.06
    gt = torch.rand(R, generator=g) * 3 + 0.3
    z = O.sample_z_with_depth(gt[:, None], tr, 32, 8, True, torch.rand(R, S, generator=g))
    sdf = torch.tanh(torch.randn(R, S, generator=g)).requires_grad_(True)
    me = types.SimpleNamespace(truncation=tr, w_sdf_fs=5, w_sdf_center=200, w_sdf_tail=10)
    l_m = RefMapper.sdf_losses(me, sdf, z, gt)
    l_m.backward()
    te = types.SimpleNamespace(truncation=tr, w_sdf_fs=10, w_sdf_center=200, w_sdf_tail=50)
    l_t = RefTracker.sdf_losses(te, sdf.detach(), z, gt)
    # empty-mask case: every sample far in front -> center/tail empty -> NaN
    z_far = torch.zeros(R, S) + 0.01
    l_nan = RefMapper.sdf_losses(me, sdf.detach(), z_far, gt)
    npz("g5_losses", gt=gt, z=z, sdf=sdf, truncation=np.float64(tr), loss_map=l_m, dsdf=sdf.grad, loss_trk=l_t,
        z_far=z_far, loss_nan=l_nan)


def g6():
    g = torch.Generator().manual_seed(6)
    # sample_pdf on its own
    B, M, K = 9, 30, 8
    zz = torch.sort(torch.rand(B, M + 2, generator=g) * 4, -1)[0]
    mid = 0.5 * (zz[:, 1:] + zz[:, :-1])                           # M+1 bin positions for M weights (Renderer.py:127-128)
    w = torch.rand(B, M, generator=g) * 0.2
    w[0] = 0                                                      # all-zero pdf row
    w[1] *= 20                                                    # cdf >> 1 (pdf is not normalised)
    u = torch.rand(B, K, generator=g)
    torch.manual_seed(61)
    s = RC.sample_pdf(mid, w, K, det=False, device=DEV)
    torch.manual_seed(61)
    u61 = torch.rand(B, K)
    # zero-depth branch through the reference renderer with the oracle grid + reference torch decoders
    cfg = make_cfg(32, 8, True)
    r = make_renderer(cfg)
    dec = RefDecoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
    enc_s, enc_c = small_grid(601), small_grid(602)
    R = 12
    ro = torch.tensor([[3.0, 1.2, 0.0]]).repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05
    rd = torch.randn(R, 3, generator=g); rd = rd / rd.norm(dim=-1, keepdim=True)
    gt = torch.rand(R, generator=g) * 2 + 0.4
    gt[::3] = 0.0                                                  # every 3rd ray has no depth
    torch.manual_seed(62)
    ret = r.render_batch_ray(([enc_s], [enc_c]), dec, rd, ro, DEV, 0.06, gt_depth=gt)
    sd = {"dec__" + k.replace(".", "__"): v for k, v in dec.state_dict().items()}
    npz("g6_zerodepth", pdf_mid=mid, pdf_w=w, pdf_u=u61, pdf_samples=s,
        rays_o=ro, rays_d=rd, gt_depth=gt, grid_s=enc_s.params, grid_c=enc_c.params, seed=62,
        z_vals=ret[5], depth=ret[2], rgb=ret[3], **sd)


def g7():
    g = torch.Generator().manual_seed(7)
    cfg = make_cfg(32, 8, True)
    r = make_renderer(cfg)
    dec = RefDecoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
    enc_s, enc_c = small_grid(701), small_grid(702)
    R = 16
    ro = (torch.tensor([[3.0, 1.2, 0.0]]).repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05).requires_grad_(True)
    rd = torch.randn(R, 3, generator=g); rd = (rd / rd.norm(dim=-1, keepdim=True)).requires_grad_(True)
    gt = torch.rand(R, generator=g) * 2 + 0.4
    torch.manual_seed(71)
    ret = r.render_batch_ray(([enc_s], [enc_c]), dec, rd, ro, DEV, 0.06, gt_depth=gt)
    term, unc, depth, rgb, sdf, z, dunc = ret
    probes = {k: torch.randn(v.shape, generator=g) for k, v in
              dict(term=term, unc=unc, depth=depth, rgb=rgb, sdf=sdf, dunc=dunc).items()}
    L = sum((probes[k] * v).sum() for k, v in dict(term=term, unc=unc, depth=depth, rgb=rgb, sdf=sdf, dunc=dunc).items())
    L.backward()
    sd = {"dec__" + k.replace(".", "__"): v for k, v in dec.state_dict().items()}
    gd = {"gdec__" + k.replace(".", "__"): v.grad for k, v in dec.named_parameters()}
    pr = {"probe_" + k: v for k, v in probes.items()}
    npz("g7_render", rays_o=ro, rays_d=rd, gt_depth=gt, grid_s=enc_s.params, grid_c=enc_c.params, seed=71,
        term=term, unc=unc, depth=depth, rgb=rgb, sdf=sdf, z_vals=z, dunc=dunc,
        g_grid_s=enc_s.params.grad, g_grid_c=enc_c.params.grad, g_rays_o=ro.grad, g_rays_d=rd.grad, **sd, **gd, **pr)


def g8():
    """Tracker.optimize_tracking (Tracker.py:149-244) with a dummy self; one iteration, both mask modes."""
    g = torch.Generator().manual_seed(8)
    H, W, fx, fy, cx, cy = 20, 24, 14.0, 14.0, 11.5, 9.5
    cfg = make_cfg(32, 8, True)
    renderer = make_renderer(cfg, H, W, fx, fy, cx, cy)
    dec = RefDecoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
    for p in dec.parameters():
        p.requires_grad_(False)
    enc_s, enc_c = small_grid(801, amp=0.3), small_grid(802, amp=0.3)
    gt_depth = (torch.rand(1, H, W, generator=g) * 1.5 + 0.5)
    gt_depth[0, 5, 5:9] = 0.0
    gt_color = torch.rand(1, H, W, 3, generator=g)
    pose0 = torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]])
    out = dict(intr=np.array([H, W, fx, fy, cx, cy]), gt_depth=gt_depth, gt_color=gt_color, pose=pose0,
               grid_s=enc_s.params, grid_c=enc_c.params, edge=np.array([2, 3]), n=48,
               **{"dec__" + k.replace(".", "__"): v for k, v in dec.state_dict().items()})
    for mode in ("original", "no_mask"):
        pose = torch.nn.Parameter(pose0.clone())
        opt = torch.optim.SGD([pose], lr=0.0)
        me = types.SimpleNamespace(cfg=cfg, hash_grids_xyz=[enc_s], c_hash_grids_xyz=[enc_c], device=DEV,
                                   H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, ignore_edge_H=2, ignore_edge_W=3,
                                   bound=BOUND, renderer=renderer, decoders=dec, truncation=0.06, t_mask_mode=mode,
                                   w_color=5, w_depth=1, w_sdf_fs=10, w_sdf_center=200, w_sdf_tail=50)
        me.sdf_losses = lambda *a, _me=me: RefTracker.sdf_losses(_me, *a)
        enc_s.params.grad = None; enc_c.params.grad = None
        torch.manual_seed(81)
        loss, unc = RefTracker.optimize_tracking(me, pose, gt_color, gt_depth, 48, opt)
        out[f"{mode}_loss"] = np.float32(loss); out[f"{mode}_unc"] = unc.detach(); out[f"{mode}_gpose"] = pose.grad.clone()
        out[f"{mode}_ggrid_s"] = enc_s.params.grad.clone()
    out["seed"] = 81
    npz("g8_tracking", **out)


def g9():
    """Mapper.optimize_mapping (Mapper.py:276-459) with a dummy self: first frame, 2 iterations incl. Adam."""
    g = torch.Generator().manual_seed(9)
    H, W, fx, fy, cx, cy = 20, 24, 14.0, 14.0, 11.5, 9.5
    cfg = make_cfg(32, 8, True)
    cfg["mapping"] = {"lr": {"decoders_lr": 0.001, "hash_grids_lr": 0.05, "c_hash_grids_lr": 0.05}}
    renderer = make_renderer(cfg, H, W, fx, fy, cx, cy)
    dec = RefDecoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
    enc_s, enc_c = small_grid(901, amp=0.3), small_grid(902, amp=0.3)
    gt_depth = (torch.rand(H, W, generator=g) * 1.5 + 0.5)
    gt_depth[5, 5:9] = 0.0
    gt_color = torch.rand(H, W, 3, generator=g)
    c2w = O.cam_pose_to_matrix(torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]]))[0]
    rays_d = RC.get_camera_rays(H, W, fx, fy, cx, cy)
    out = dict(intr=np.array([H, W, fx, fy, cx, cy]), gt_depth=gt_depth, gt_color=gt_color, c2w=c2w,
               grid_s0=enc_s.params.detach().clone(), grid_c0=enc_c.params.detach().clone(), pixels=40, iters=2,
               **{"dec0__" + k.replace(".", "__"): v.clone() for k, v in dec.state_dict().items()})
    me = types.SimpleNamespace(cfg=cfg, hash_grids_xyz=[enc_s], c_hash_grids_xyz=[enc_c], device=DEV,
                               H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, bound=BOUND, renderer=renderer, decoders=dec,
                               truncation=0.06, m_mask_mode="original", w_color=5, w_depth=0.1, w_sdf_fs=5,
                               w_sdf_center=200, w_sdf_tail=10, keyframe_selection_method="global",
                               keyframe_dict=[], keyframe_list=[], mapping_pixels=40, joint_opt=False,
                               no_vis_on_first_frame=True, tracking_back=torch.zeros(1).int(),
                               mapping_window_size=20)
    me.sdf_losses = lambda *a: RefMapper.sdf_losses(me, *a)
    me.create_optimizer = lambda c, f: RefMapper.create_optimizer(me, c, f)
    torch.manual_seed(91)
    RefMapper.optimize_mapping(me, 2, 5.0, 0, gt_color, gt_depth, c2w, [], [], c2w, rays_d)
    out.update(seed=91, lr_factor=5.0, grid_s1=enc_s.params.detach(), grid_c1=enc_c.params.detach(),
               **{"dec1__" + k.replace(".", "__"): v for k, v in dec.state_dict().items()})
    npz("g9_mapping", **out)


def g14():
    """Mapper.optimize_mapping (Mapper.py:276-459) with joint_opt = True and a dummy self: the window's poses are a fourth Adam group."""
    H, W, fx, fy, cx, cy = 24, 32, 18.0, 18.0, 15.5, 11.5
    cfg = make_cfg(32, 8, True)
    cfg["mapping"] = {"lr": {"decoders_lr": 0.001, "hash_grids_lr": 0.05, "c_hash_grids_lr": 0.05}}
    renderer = make_renderer(cfg, H, W, fx, fy, cx, cy)
    torch.manual_seed(1400)              # the decoders' initial weights come from the global generator: fixed here, whatever ran before
    dec0 = RefDecoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
    sd0 = {k: v.clone() for k, v in dec0.state_dict().items()}
    gs0, gc0 = small_grid(1401, amp=0.3).params.detach().clone(), small_grid(1402, amp=0.3).params.detach().clone()
    rays_d = RC.get_camera_rays(H, W, fx, fy, cx, cy)
    out = dict(intr=np.array([H, W, fx, fy, cx, cy]), grid_s0=gs0, grid_c0=gc0, cam_lr=0.001, lr_factor=1.0,
               **{"dec0__" + k.replace(".", "__"): v for k, v in sd0.items()})
    base = torch.tensor([0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0])
    for tag, n_kf, picks, pixels, seed in (("w6", 5, [0, 1, 2], 120, 141), ("w12x", 22, [0, 2, 4, 6, 8, 10, 12, 14, 16], 120, 142)):
        g = torch.Generator().manual_seed(seed)
        n_pool = int(H * W * 0.1)
        kd, kl = [], []
        for k in range(n_kf):
            pose = base + torch.cat([torch.randn(4, generator=g) * 0.05, torch.randn(3, generator=g) * 0.1])
            c2w = O.cam_pose_to_matrix(pose[None])[0]
            depth = torch.rand(H * W, generator=g) * 1.5 + 0.5
            depth[3::37] = 50.0                                              # beyond the scene box: dropped by the pre-filter
            color = torch.rand(H * W, 3, generator=g)
            ind = torch.randperm(H * W, generator=g)[:n_pool]
            kd.append({"gt_c2w": c2w.clone(), "idx": 4 * k, "color": color[ind], "depth": depth[ind], "est_c2w": c2w.clone(),
                       "rays_d": rays_d.reshape(-1, 3)[ind]})
            kl.append(4 * k)
        cur_depth = torch.rand(H, W, generator=g) * 1.5 + 0.5
        cur_depth[7, 3:9] = 50.0
        cur_color = torch.rand(H, W, 3, generator=g)
        cur_c2w = O.cam_pose_to_matrix((base + torch.cat([torch.randn(4, generator=g) * 0.05, torch.randn(3, generator=g) * 0.1]))[None])[0]
        frames = sorted(picks + [n_kf - 1, n_kf - 2])
        out.update({f"{tag}_n_kf": n_kf, f"{tag}_frames": np.array(frames), f"{tag}_pixels": pixels, f"{tag}_seed": seed,
                    f"{tag}_kf_c2w": torch.stack([d["est_c2w"] for d in kd]), f"{tag}_kf_depth": torch.stack([d["depth"] for d in kd]),
                    f"{tag}_kf_color": torch.stack([d["color"] for d in kd]), f"{tag}_kf_dirs": torch.stack([d["rays_d"] for d in kd]),
                    f"{tag}_cur_depth": cur_depth, f"{tag}_cur_color": cur_color, f"{tag}_cur_c2w": cur_c2w})
        for iters in (1, 2):
            dec = RefDecoders(cfg, c_dim=32, truncation=0.06, learnable_beta=True)
            dec.load_state_dict(sd0)
            enc_s, enc_c = small_grid(1401, amp=0.3), small_grid(1402, amp=0.3)
            kd_run = [dict(d, est_c2w=d["est_c2w"].clone()) for d in kd]
            me = types.SimpleNamespace(cfg=cfg, hash_grids_xyz=[enc_s], c_hash_grids_xyz=[enc_c], device=DEV,
                                       H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, bound=BOUND, renderer=renderer, decoders=dec,
                                       truncation=0.06, m_mask_mode="original", w_color=5, w_depth=0.1, w_sdf_fs=5,
                                       w_sdf_center=200, w_sdf_tail=10, keyframe_selection_method="global",
                                       keyframe_dict=kd_run, keyframe_list=kl, mapping_pixels=pixels, joint_opt=True, joint_opt_cam_lr=0.001,
                                       no_vis_on_first_frame=True, tracking_back=torch.zeros(1).int(), mapping_window_size=20,
                                       visualizer=MagicMock())
            me.sdf_losses = lambda *a, _me=me: RefMapper.sdf_losses(_me, *a)
            me.create_optimizer = lambda c, f, _me=me: RefMapper.create_optimizer(_me, c, f)
            me.keyframe_selection_LC = lambda *a, _p=picks: list(_p)       # SLAM policy (pinned by g10), kept out of the random stream
            torch.manual_seed(seed)
            new_c2w = RefMapper.optimize_mapping(me, iters, 1.0, 4 * n_kf, cur_color, cur_depth, cur_c2w, kd_run, kl, cur_c2w.clone(), rays_d)
            cam = me.optimizer.param_groups[3]["params"][0]
            pre = f"{tag}_i{iters}_"
            out.update({pre + "poses": cam.detach().clone(), pre + "cur_c2w": new_c2w.detach().clone(),
                        pre + "kf_c2w": torch.stack([d["est_c2w"].detach() for d in kd_run])})
            if iters == 1:                                                   # the gradients of the (only) iteration
                out.update({pre + "g_poses": cam.grad.clone(), pre + "g_beta": dec.beta.grad.clone()})
                if tag == "w6":
                    out.update({pre + "g_grid_s": enc_s.params.grad.clone(), pre + "g_grid_c": enc_c.params.grad.clone()})
            else:                                                            # the state after two optimiser steps
                out.update({pre + "grid_s": enc_s.params.detach().clone(), pre + "grid_c": enc_c.params.detach().clone(),
                            **{pre + "dec__" + k.replace(".", "__"): v.clone() for k, v in dec.state_dict().items()}})
    npz("g14_mapping_joint", **out)



def g10():
    """Mapper.keyframe_selection_LC (Mapper.py:177-274) with a dummy self; three situations."""
    g = torch.Generator().manual_seed(10)
    H, W, fx, fy, cx, cy = 60, 80, 50.0, 50.0, 39.5, 29.5
    gt_depth = torch.rand(H, W, generator=g) * 1.5 + 1.0
    gt_depth[10, 5:25] = 0.0
    gt_color = torch.rand(H, W, 3, generator=g)
    base = O.cam_pose_to_matrix(torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]]))[0]
    n_frames = 140
    est = torch.zeros(n_frames, 4, 4)
    # keyframes every 10 frames: the camera drifts away and turns, then comes back to the start (loop) at frame 130
    kfs = list(range(0, 130, 10))
    for k in kfs:
        t = k / 120.0
        ang = 2.5 * np.sin(np.pi * t)                                        # turns away, returns at t = 1
        q = torch.tensor([[np.cos(ang / 2), 0.0, np.sin(ang / 2), 0.0, 0.0, 0.0, 0.0]], dtype=torch.float32)
        rot = O.cam_pose_to_matrix(q)[0]
        c = base.clone(); c[:3, :3] = base[:3, :3] @ rot[:3, :3]; c[:3, 3] = base[:3, 3] + torch.tensor([0.8 * np.sin(np.pi * t), 0.3 * t * (1 - t), 0.0], dtype=torch.float32)
        est[k] = c
    out = dict(intr=np.array([H, W, fx, fy, cx, cy]), gt_depth=gt_depth, gt_color=gt_color, c2w=base, est=est, keyframe_list=np.array(kfs))
    for tag, idx, tb, seed in (("plain", 35, 0, 101), ("loop", 135, 0, 102), ("back", 35, 1, 103)):
        me = types.SimpleNamespace(device=DEV, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, estimate_c2w_list=est, LC=True,
                                   keyframe_list=kfs if tag != "plain" else kfs[:6], LC_cnt=torch.zeros(1).int(),
                                   tracking_back=torch.tensor([tb]).int(), activated_mapping_mode=True)
        torch.manual_seed(seed)
        num = len(me.keyframe_list) - 2
        sel = RefMapper.keyframe_selection_LC(me, num, idx, gt_color, gt_depth, base, 5)
        out.update({f"{tag}_idx": idx, f"{tag}_tb": tb, f"{tag}_seed": seed, f"{tag}_n_kf": len(me.keyframe_list),
                    f"{tag}_sel": np.array([int(v) for v in sel], dtype=np.int64), f"{tag}_lc": int(me.LC_cnt[0])})
    npz("g10_keyframes", **out)


def g11():
    """src/utils/datasets.py pose parsers + TUM association on generated text files (the files' text is part of the fixture)."""
    import tempfile
    if not hasattr(np, "unicode_"):
        np.unicode_ = np.str_                  # alias removed in numpy 2 (datasets.py:242 still names it)
    import src.utils.datasets as RD
    rng = np.random.default_rng(11)

    def rand_pose():
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4); M[:3, :3] = R; M[:3, 3] = rng.normal(size=3) * 2
        return M, q

    cam = dict(H=48, W=64, fx=50.0, fy=50.0, cx=31.5, cy=23.5, png_depth_scale=1000.0, crop_edge=0)
    out, files = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        def put(rel, text):
            path = os.path.join(tmp, rel)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(text)
            files[rel] = text

        def run(cls, name, folder, extra=None):
            cfg = {"dataset": name, "cam": dict(cam), "data": {"input_folder": os.path.join(tmp, folder)}}
            cfg["data"].update(extra or {})
            return cls(cfg, types.SimpleNamespace(input_folder=None), 1.0, device="cpu")

        # Replica: 5 frames, traj.txt one matrix per line (6 lines: the reader stops at n_img)
        n = 5
        for k in range(n):
            put(f"replica/results/frame{k:06d}.jpg", ""); put(f"replica/results/depth{k:06d}.png", "")
        put("replica/traj.txt", "".join(" ".join(f"{v:.17e}" for v in rand_pose()[0].reshape(-1)) + "\n" for _ in range(n + 1)))
        ds = run(RD.Replica, "replica", "replica")
        out["replica_poses"] = torch.stack(ds.poses)
        out["replica_color"] = np.array([os.path.basename(p) for p in ds.color_paths])
        # ScanNet: numeric ordering (2 < 10), pose files of 4 single-space rows
        for k in (0, 2, 10, 1):
            put(f"scannet/color/{k}.jpg", ""); put(f"scannet/depth/{k}.png", "")
            put(f"scannet/pose/{k}.txt", "".join(" ".join(f"{v:.9f}" for v in row) + "\n" for row in rand_pose()[0]))
        ds = run(RD.ScanNet, "scannet", "scannet")
        out["scannet_poses"] = torch.stack(ds.poses)
        out["scannet_color"] = np.array([os.path.basename(p) for p in ds.color_paths])
        # Azure: trajectory.log, 5-line records
        for k in range(3):
            put(f"azure/color/{k:04d}.jpg", ""); put(f"azure/depth/{k:04d}.png", "")
        put("azure/scene/trajectory.log", "".join(f"{k} {k} {k + 1}\n" + "".join(" ".join(f"{v:.8f}" for v in row) + "\n" for row in rand_pose()[0])
                                                 for k in range(3)))
        ds = run(RD.Azure, "azure", "azure")
        out["azure_poses"] = torch.stack(ds.poses)
        # RGBDataset: natural file order, poses.txt with a nan block, no axis flip
        for k in (1, 2, 10):
            put(f"rgbd/images/img{k}.png", ""); put(f"rgbd/depth_gt/d{k}.png", "")
        blocks = ["".join(" ".join(f"{v:.7f}" for v in row) + "\n" for row in rand_pose()[0]) for _ in range(3)]
        blocks[1] = "nan nan nan nan\n" * 4
        put("rgbd/poses.txt", "".join(blocks))
        ds = run(RD.RGBDataset, "systheticrgbd", "rgbd", {"depth_folder": "depth"})
        out["rgbd_poses"] = torch.stack([torch.as_tensor(np.asarray(p, dtype=np.float32)) for p in ds.poses])
        out["rgbd_color"] = np.array([os.path.basename(p) for p in ds.color_paths])
        out["rgbd_depth"] = np.array([os.path.basename(p) for p in ds.depth_paths])
        # TUM: jittered 30 Hz colour stamps (some closer than 1/32 s), depth with a hole, 100 Hz poses with a hole
        t0 = 1305031102.175304
        t_img = t0 + np.cumsum(np.concatenate([[0.0], 0.0333 + rng.uniform(-0.006, 0.006, size=59)]))
        t_dep = np.array([t for k, t in enumerate(t_img + rng.uniform(-0.02, 0.02, size=60)) if not (20 <= k < 26)])
        t_pos = np.array([t for t in np.arange(t_img[0] - 0.05, t_img[-1] + 0.05, 0.01) if not (t_img[40] - 0.02 < t < t_img[44] + 0.1)])
        put("tum/rgb.txt", "# color images\n# file: 'x.bag'\n# timestamp filename\n" + "".join(f"{t:.6f} rgb/{t:.6f}.png\n" for t in t_img))
        put("tum/depth.txt", "# depth maps\n# file: 'x.bag'\n# timestamp filename\n" + "".join(f"{t:.6f} depth/{t:.6f}.png\n" for t in t_dep))
        rows = []
        for t in t_pos:
            M, q = rand_pose()
            rows.append(f"{t:.4f} " + " ".join(f"{v:.4f}" for v in list(M[:3, 3]) + list(q)) + "\n")
        put("tum/groundtruth.txt", "# ground truth trajectory\n# file: 'x.bag'\n# timestamp tx ty tz qx qy qz qw\n" + "".join(rows))
        ds = run(RD.TUM_RGBD, "tumrgbd", "tum")
        out["tum_poses"] = torch.stack(ds.poses)
        out["tum_color"] = np.array([os.path.relpath(p, os.path.join(tmp, "tum")) for p in ds.color_paths])
        out["tum_depth"] = np.array([os.path.relpath(p, os.path.join(tmp, "tum")) for p in ds.depth_paths])
    out["cam"] = np.array([cam[k] for k in ("H", "W", "fx", "fy", "cx", "cy", "png_depth_scale", "crop_edge")], dtype=np.float64)
    out["file_names"] = np.array(list(files.keys()))
    out["file_texts"] = np.array(list(files.values()))
    npz("g11_datasets", **out)


def g12():
    """src/tools/eval_ate.py:169-236,270-281,380-552 on generated trajectories (plots go to a temporary folder)."""
    import tempfile
    import contextlib, io
    os.environ.setdefault("MPLBACKEND", "Agg")
    import src.tools.eval_ate as E
    # pytorch3d is absent (see the header): only the translations enter the ATE, so any quaternion serves in convert_poses
    # (for THIS generator only: g14 behind it needs the real helper -- r5: the patch used to stay in place for the rest of the process)
    RC.matrix_to_quaternion = lambda R: torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(R.shape[0], 1)
    try:
        _g12_body(E)
    finally:
        RC.matrix_to_quaternion = O.matrix_to_quaternion


def _g12_body(E):
    import tempfile
    import contextlib, io
    rng = np.random.default_rng(12)
    out = {}
    n = 60
    t = np.linspace(0, 1, n)
    gt = np.stack([2 * np.sin(2 * np.pi * t), 1.5 * np.cos(2 * np.pi * t) + 0.3 * t, 0.4 * np.sin(4 * np.pi * t)], 0)      # [3,n]
    ang = 0.3
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    Rx = np.array([[1, 0, 0], [0, np.cos(0.2), -np.sin(0.2)], [0, np.sin(0.2), np.cos(0.2)]])
    cases = {
        "noisy": Rz @ Rx @ gt + np.array([[0.5], [-0.2], [0.1]]) + rng.normal(size=gt.shape) * 0.01,
        "drift": gt + np.cumsum(rng.normal(size=gt.shape) * 0.002, axis=1),
        "planar": np.concatenate([gt[:2] * 0.98 + rng.normal(size=(2, n)) * 0.005, np.zeros((1, n))], 0),       # rank-2 data: the det() branch
    }
    gts = {"noisy": gt, "drift": gt, "planar": np.concatenate([gt[:2], np.zeros((1, n))], 0) * np.array([[1.0], [-1.0], [1.0]])}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        for name, est in cases.items():
            g_ = gts[name]
            rot, trans, err = E.align(np.matrix(est), np.matrix(g_))
            out[f"{name}_gt"], out[f"{name}_est"] = g_, est
            out[f"{name}_rot"], out[f"{name}_trans"], out[f"{name}_err"] = np.asarray(rot), np.asarray(trans), np.asarray(err)
            first = {i: g_[:, i] for i in range(n)}
            second = {i: est[:, i] for i in range(n)}
            for pa in (False, True):
                te, res = E.evaluate_ate(first, second, os.path.join(tmp, f"{name}_{int(pa)}.png"), _args=[], pose_alignment=pa)
                out[f"{name}_te{int(pa)}"] = np.asarray(te)
                out[f"{name}_res{int(pa)}"] = np.array([res["compared_pose_pairs"], res["error.rmse"], res["error.mean"], res["error.median"],
                                                        res["error.std"], res["error.max"]], dtype=np.float64)
        # associate: unequal stamps, an offset, one stamp out of reach, a contested match
        a = {0.00: 0, 0.10: 1, 0.20: 2, 0.31: 3, 0.50: 4}
        b = {0.012: 0, 0.095: 1, 0.205: 2, 0.214: 3, 0.299: 4, 0.60: 5}
        out["assoc_a"], out["assoc_b"] = np.array(list(a.keys())), np.array(list(b.keys()))
        out["assoc_m"] = np.array(E.associate(a, b, 0.0, 0.02))
        out["assoc_m_off"] = np.array(E.associate(a, b, -0.01, 0.02))
        # pose_evaluation: c2w lists with an invalid given pose (ScanNet) and scale 2
        N = 12
        c2w_gt = O.cam_pose_to_matrix(torch.cat([torch.nn.functional.normalize(torch.as_tensor(rng.normal(size=(N, 4)), dtype=torch.float32), dim=-1),
                                                 torch.as_tensor(gts["noisy"][:, :N].T, dtype=torch.float32)], -1))
        c2w_est = c2w_gt.clone()
        c2w_est[:, :3, 3] += torch.as_tensor(rng.normal(size=(N, 3)) * 0.02, dtype=torch.float32)
        c2w_gt[5, 0, 0] = float("inf"); c2w_gt[8, 1, 3] = float("nan")
        out["pe_gt"], out["pe_est"] = c2w_gt.clone(), c2w_est.clone()
        te, res = E.pose_evaluation(c2w_gt.clone(), c2w_est.clone(), torch.zeros(N), os.path.join(tmp, "pe.png"), 2.0, False)
        out["pe_te"] = np.asarray(te)
        out["pe_res"] = np.array([res["compared_pose_pairs"], res["error.rmse"], res["error.mean"], res["error.median"], res["error.std"],
                                  res["error.max"]], dtype=np.float64)
    npz("g12_ate", **out)


def g13():
    """src/UNISLAM.py:168-218,241 with a dummy self + src/config.py:21-53 on the reference's own config files."""
    from src.UNISLAM import UNISLAM as RefU
    from src import config as RCfg
    out = {}
    cwd = os.getcwd()
    os.chdir(REF)                                  # inherit_from paths are relative to the reference root
    try:
        for tag, path in (("room0", "configs/Replica/room0.yaml"), ("scene0000", "configs/ScanNet/scene0000.yaml"),
                          ("fr1_desk", "configs/TUM_RGBD/freiburg1_desk.yaml")):
            cfg = RCfg.load_config(path, "configs/UNISLAM.yaml")
            cam = cfg["cam"]
            me = types.SimpleNamespace(cfg=cfg, scale=cfg["scale"], shared_decoders=types.SimpleNamespace(),
                                       H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
            me.get_resolution = lambda c, me=me: RefU.get_resolution(me, c)
            RefU.update_cam(me)
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                RefU.load_bound(me, cfg)
            out[f"{tag}_bound_in"] = np.array(cfg["mapping"]["bound"], dtype=np.float64)
            out[f"{tag}_bound"] = me.bound
            out[f"{tag}_res"] = np.array([me.resolution_sdf, me.resolution_color])
            out[f"{tag}_cam_in"] = np.array([cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["crop_edge"]] +
                                            list(cam.get("crop_size", [0, 0])), dtype=np.float64)
            out[f"{tag}_cam"] = np.array([me.H, me.W, me.fx, me.fy, me.cx, me.cy], dtype=np.float64)
            out[f"{tag}_voxel"] = np.array([cfg["grid"]["voxel_sdf"], cfg["grid"]["voxel_color"], cfg["planes_res"]["bound_dividable"], cfg["scale"]])
            # the per_level_scale expression of get_encoder (UNISLAM.py:241), evaluated as written there
            n_levels = 16
            out[f"{tag}_pls"] = np.array([np.exp2(np.log2(r / n_levels) / (n_levels - 1)) for r in (me.resolution_sdf, me.resolution_color)])
            # merged configuration: a few leaves from each level of the inheritance chain
            out[f"{tag}_cfg"] = np.array([cfg["tracking"]["lr_T"], cfg["tracking"]["iters"], cfg["mapping"]["pixels"], cfg["mapping"]["iters"],
                                          cfg["mapping"]["lr"]["hash_grids_lr"], cfg["mapping"]["lr"]["decoders_lr"], cfg["grid"]["hash_size_sdf"],
                                          cfg["grid"]["hash_size_color"], cfg["rendering"]["n_stratified"], cfg["mapping"]["w_sdf_fs"],
                                          cfg["tracking"]["w_sdf_tail"], cfg["tracking"]["uncertainty_ts"], float(cfg["grid"]["tcnn_network"]),
                                          float(cfg["rendering"]["learnable_beta"])], dtype=np.float64)
    finally:
        os.chdir(cwd)
    npz("g13_scene", **out)


from g15_settings import G15, G16  # noqa: E402  (shared with tests/test_gpu_slam.py: the HIP drivers run with the same numbers)


class _DrawLog:
    """Records every draw the reference's loop takes from torch's global CPU generator -- torch.randint (pixel indices: src/common.py:116,155),
    torch.rand (z jitter: src/utils/Renderer.py:55; sample_pdf: src/common.py:64), torch.randperm (keyframe pools: src/Mapper.py:335,518; the
    tracking-back draw: :257) -- WITHOUT touching the stream: per call its kind, two size numbers, the f64 sum of what was drawn and the frame it
    belongs to.  A replay (tests/test_gpu_slam.py: a torch.Generator set to the stored state, consumed in the same order by the HIP drivers)
    is checked against this log call by call."""
    KINDS = ("randint", "rand", "randperm")

    def __init__(self):
        self.kind, self.a, self.b, self.sum, self.frame, self.cur = [], [], [], [], [], 0
        self.loss, self.loss_frame = [], []                              # every scalar the loop calls .backward() on, in order (the iterations' losses)

    def __enter__(self):
        self._orig = {k: getattr(torch, k) for k in self.KINDS}
        for k in self.KINDS:
            setattr(torch, k, self._wrap(k))
        self._backward = torch.Tensor.backward
        log, orig_backward = self, self._backward

        def backward(t, *a, **kw):
            log.loss.append(float(t.detach())); log.loss_frame.append(log.cur)
            return orig_backward(t, *a, **kw)
        torch.Tensor.backward = backward
        return self

    def __exit__(self, *a):
        for k, f in self._orig.items():
            setattr(torch, k, f)
        torch.Tensor.backward = self._backward
        return False

    def _wrap(self, k):
        orig = self._orig[k]

        def f(*args, **kw):
            out = orig(*args, **kw)
            if k == "randint":
                a, b = int(args[0]), out.numel()
            elif k == "rand":
                a, b = (int(out.shape[0]), int(out.shape[1])) if out.dim() == 2 else (out.numel(), 1)
            else:
                a, b = int(args[0]), 0
            self.kind.append(self.KINDS.index(k)); self.a.append(a); self.b.append(b)
            self.sum.append(float(out.double().sum())); self.frame.append(self.cur)
            return out
        return f

    def arrays(self):
        return dict(draw_kind=np.array(self.kind, dtype=np.uint8), draw_a=np.array(self.a, dtype=np.int32), draw_b=np.array(self.b, dtype=np.int32),
                    draw_sum=np.array(self.sum, dtype=np.float64), draw_frame=np.array(self.frame, dtype=np.int32),
                    loss_log=np.array(self.loss, dtype=np.float32), loss_frame=np.array(self.loss_frame, dtype=np.int32))


def _ref_loop(P):
    """
    The reference's LOOP over a sequence: Tracker.run's body (Tracker.py:296-366: constant-speed prediction, a fresh pose Adam with
    betas (0.5, 0.999), optimize_tracking per iteration, the minimum-loss pose, the uncertainty-triggered doubling of the iteration counts
    and the tracking-back flag) and Mapper.run's body (Mapper.py:494-533: first-frame factor / iterations, joint_opt from the fifth keyframe,
    optimize_mapping with the REAL keyframe_selection_LC, the new keyframe's 10 % pool) alternating frame by frame as the two processes do
    through their wait loops.  optimize_tracking / optimize_mapping / keyframe_selection_LC / create_optimizer / sdf_losses are the
    reference's methods on dummy selves; the encoders are the oracle's CPU hash grids, the frames come from the build's analytic
    SyntheticRoom (its parameters are the fixture's inputs).  Returns the fixture's arrays: the estimated trajectory, the keyframe list,
    per-frame errors and flags, iteration counts, the windows keyframe_selection_LC chose, the generator state at the loop's start and
    the log of every draw (_DrawLog).
    """
    import copy
    sys.path.insert(0, ROOT)
    import unislam_amd  # noqa: F401  (the analytic scene only; nothing of the HIP path runs here)
    from unislam_amd.synthetic import SyntheticRoom
    T_, M_ = P["tracking"], P["mapping"]
    torch.manual_seed(P["seed"])
    NF, H, W = P["n_frames"], P["H"], P["W"]
    room = SyntheticRoom(n_frames=NF, H=H, W=W, fov_deg=P["fov_deg"], device="cpu", tex_freq=P["tex_freq"])
    fx, fy, cx, cy = room.fx, room.fy, room.cx, room.cy
    bound = O.load_bound(P["room_bound"])
    res = int((bound[:, 1] - bound[:, 0]).max() / P["voxel"])
    mk = lambda l2, seed: O.HashGridOracle(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                               "base_resolution": 16, "per_level_scale": O.per_level_scale(res)})
    enc_s, enc_c = mk(P["log2T"][0], 1), mk(P["log2T"][1], 2)
    gi = torch.Generator().manual_seed(P["seed"] + 1)
    with torch.no_grad():                                                   # tcnn's initial range, from a generator of this fixture's own
        enc_s.params.copy_((torch.rand(enc_s.params.shape, generator=gi) * 2 - 1) * 1e-4)
        enc_c.params.copy_((torch.rand(enc_c.params.shape, generator=gi) * 2 - 1) * 1e-4)
    table0_sums = (float(enc_s.params.detach().double().sum()), float(enc_c.params.detach().double().sum()))    # (a replay restates this draw)
    cfg = make_cfg(P["n_stratified"], P["n_importance"], True)
    cfg["mapping"] = {"lr": {k: M_[k] for k in ("decoders_lr", "hash_grids_lr", "c_hash_grids_lr")}}
    u = types.SimpleNamespace(bound=bound, device=DEV, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy)
    renderer = RefRenderer(cfg, u)
    dec = RefDecoders(cfg, c_dim=32, truncation=P["truncation"], learnable_beta=True)
    dec.bound = bound
    dec0 = {k: v.clone() for k, v in dec.state_dict().items()}
    dec_t = copy.deepcopy(dec)                                              # Tracker.py:106-111: the tracker's own copy, frozen
    for p_ in dec_t.parameters():
        p_.requires_grad_(False)
    est = torch.zeros(NF, 4, 4)
    gts = room.poses.clone()
    tracking_back = torch.zeros(1).int()
    common = dict(cfg=cfg, hash_grids_xyz=[enc_s], c_hash_grids_xyz=[enc_c], device=DEV, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, bound=bound,
                  renderer=renderer, truncation=P["truncation"], estimate_c2w_list=est, tracking_back=tracking_back)
    trk = types.SimpleNamespace(decoders=dec_t, ignore_edge_H=T_["ignore_edge_H"], ignore_edge_W=T_["ignore_edge_W"], t_mask_mode="original",
                                **{k: T_[k] for k in ("w_color", "w_depth", "w_sdf_fs", "w_sdf_center", "w_sdf_tail")}, **common)
    trk.sdf_losses = lambda *a: RefTracker.sdf_losses(trk, *a)
    mp = types.SimpleNamespace(decoders=dec, m_mask_mode="original", keyframe_selection_method="global", keyframe_dict=[], keyframe_list=[],
                               mapping_pixels=M_["pixels"], joint_opt=False, joint_opt_cam_lr=M_["joint_opt_cam_lr"], no_vis_on_first_frame=True,
                               mapping_window_size=M_["mapping_window_size"], visualizer=MagicMock(), LC=M_["LC"], LC_cnt=torch.zeros(1).int(),
                               activated_mapping_mode=T_["activated_mapping_mode"],
                               **{k: M_[k] for k in ("w_color", "w_depth", "w_sdf_fs", "w_sdf_center", "w_sdf_tail")}, **common)
    mp.sdf_losses = lambda *a: RefMapper.sdf_losses(mp, *a)
    mp.create_optimizer = lambda c, f: RefMapper.create_optimizer(mp, c, f)
    selections = []

    def select(*a, **k):                                                    # (the reference's method; what it returned is kept for the fixture)
        sel = RefMapper.keyframe_selection_LC(mp, *a, **k)
        selections.append([int(x) for x in sel])
        return sel
    mp.keyframe_selection_LC = select
    num_cam_iters, m_iters = T_["iters"], M_["iters"]
    init_phase = True
    tb_flags, joint_flags, window_sizes, track_iters, map_iters, mapped, unc_mean, unc_evals = [], [], [], [], [], [], [], []
    rng_state = torch.get_rng_state().clone()                                # the global generator as the loop finds it
    snaps, pool_idx = {}, []
    log = _DrawLog()
    with log:
        for idx in range(NF):
            log.cur = idx
            _, color, depth, gt_c2w, rays_d = room[idx]
            if idx in P.get("snapshots", ()):
                # the loop's whole state as frame idx finds it: a replay can START here (tests: one frame from the reference's own state)
                tag = f"snap{idx}__"
                snaps.update({tag + "table_sdf": enc_s.params.detach().clone(), tag + "table_color": enc_c.params.detach().clone(),
                              tag + "rng_state": torch.get_rng_state().clone(), tag + "draw_pos": len(log.kind), tag + "loss_pos": len(log.loss),
                              tag + "num_cam_iters": num_cam_iters, tag + "m_iters": m_iters, tag + "tracking_back": int(tracking_back[0]),
                              tag + "n_keyframes": len(mp.keyframe_list), tag + "n_mapped": len(mapped),
                              tag + "kf_est_c2w": torch.stack([d["est_c2w"].detach() for d in mp.keyframe_dict])})
                snaps.update({tag + "dec__" + k.replace(".", "__"): v.detach().clone() for k, v in dec.state_dict().items()})
            # ---- Tracker.run body (Tracker.py:296-366)
            dec_t.load_state_dict(dec.state_dict())                             # update_params_from_mapping (:246-258): the grids are shared objects
            n_it, w_frame = 0, float("nan")
            if idx == 0:
                c2w = gt_c2w.clone()
            else:
                pre_c2w = est[idx - 1].unsqueeze(0)
                if T_["const_speed_assumption"] and idx - 2 >= 0:
                    pre_poses = RC.matrix_to_cam_pose(torch.stack([est[idx - 2], pre_c2w.squeeze(0)], dim=0))
                    cam_pose = 2 * pre_poses[1:] - pre_poses[0:1]
                else:
                    cam_pose = RC.matrix_to_cam_pose(pre_c2w)
                T = torch.nn.Parameter(cam_pose[:, -3:].clone())
                R = torch.nn.Parameter(cam_pose[:, :4].clone())
                opt = torch.optim.Adam([{"params": [T], "lr": T_["lr_T"], "betas": (0.5, 0.999)}, {"params": [R], "lr": T_["lr_R"], "betas": (0.5, 0.999)}])
                current_min_loss, cam_iter = float("inf"), 0
                while cam_iter < num_cam_iters:
                    cam_pose = torch.cat([R, T], -1)
                    loss, rendered_weights = RefTracker.optimize_tracking(trk, cam_pose, color[None], depth[None], T_["pixels"], opt)
                    if loss < current_min_loss:
                        current_min_loss, candidate = loss, cam_pose.clone().detach()
                    cam_iter += 1
                    if cam_iter == num_cam_iters - 1:
                        w = rendered_weights.detach().mean()
                        w_frame = float(w)
                        unc_evals.append((idx, cam_iter, w_frame))          # (a frame that doubles its count is evaluated again at the new count - 1)
                        if T_["activated_mapping_mode"] and w > T_["uncertainty_ts"]:
                            num_cam_iters, m_iters = T_["iters"] * 2, M_["iters"] * 2
                            tracking_back[0] = 1
                        else:
                            num_cam_iters, m_iters = T_["iters"], M_["iters"]
                            tracking_back[0] = 0
                n_it = cam_iter
                c2w = RC.cam_pose_to_matrix(candidate).squeeze(0)
            est[idx] = c2w.detach().clone()
            tb_flags.append(int(tracking_back[0])); track_iters.append(n_it); unc_mean.append(w_frame)
            # ---- Mapper.run body (Mapper.py:494-533)
            if idx % M_["every_frame"] == 0 or int(tracking_back[0]) == 1 or idx == NF - 1:
                cur_c2w = est[idx]
                lr_factor = M_["lr_first_factor"] if init_phase else M_["lr_factor"]
                iters = M_["iters_first"] if init_phase else m_iters
                mp.joint_opt = (len(mp.keyframe_list) > 4) and M_["joint_opt"]
                joint_flags.append(int(mp.joint_opt))
                window_sizes.append(len(mp.keyframe_list))
                mapped.append(idx); map_iters.append(int(iters))
                if len(mp.keyframe_dict) == 0:
                    selections.append([])                                   # (no selection call on an empty keyframe list, Mapper.py:303-304)
                cur_c2w = RefMapper.optimize_mapping(mp, iters, lr_factor, idx, color, depth, gt_c2w, mp.keyframe_dict, mp.keyframe_list, cur_c2w, rays_d)
                if mp.joint_opt:
                    est[idx] = cur_c2w.detach()
                if idx % M_["keyframe_every"] == 0 or int(tracking_back[0]) == 1:
                    mp.keyframe_list.append(idx)
                    n_save = int(H * W * 0.1)
                    ind = torch.randperm(H * W)[:n_save]
                    pool_idx.append(ind.clone())
                    mp.keyframe_dict.append({"gt_c2w": gt_c2w, "idx": idx, "color": color.reshape(-1, 3)[ind], "depth": depth.reshape(-1)[ind],
                                             "est_c2w": cur_c2w.detach().clone(), "rays_d": rays_d.reshape(-1, 3)[ind]})
                init_phase = False
            if idx in P.get("snapshots", ()):
                snaps[f"snap{idx}__kf_est_c2w_after"] = torch.stack([d["est_c2w"].detach() for d in mp.keyframe_dict])
    err = (est[:, :3, 3] - gts[:, :3, 3]).norm(dim=-1)
    _, c0, d0, _, _ = room[0]
    sel_off = np.cumsum([0] + [len(s_) for s_ in selections])
    out = dict(gt_c2w=gts, est_c2w=est, keyframe_list=np.array(mp.keyframe_list), tracking_back=np.array(tb_flags),
               joint_opt=np.array(joint_flags), keyframes_before_mapping=np.array(window_sizes), lc_cnt=int(mp.LC_cnt[0]), err_m=err,
               ate_rmse_m=float(err.pow(2).mean().sqrt()), intr=np.array([H, W, fx, fy, cx, cy]), frame0_depth_row=d0[H // 2], frame0_color_row=c0[H // 2],
               res=res, kf_est_c2w=torch.stack([d["est_c2w"] for d in mp.keyframe_dict]),
               **{"dec0__" + k.replace(".", "__"): v for k, v in dec0.items()},
               **{"dec1__" + k.replace(".", "__"): v.detach() for k, v in dec.state_dict().items()})
    # (r6) what a draw-for-draw replay needs and checks: the generator's state at the loop's start, the draw log, the loop's decisions
    out.update(rng_state=rng_state.numpy(), track_iters=np.array(track_iters), mapped_frames=np.array(mapped), map_iters=np.array(map_iters),
               unc_mean=np.array(unc_mean, dtype=np.float64), unc_evals=np.array(unc_evals, dtype=np.float64).reshape(-1, 3), selected_flat=np.array([x for s_ in selections for x in s_], dtype=np.int32),
               selected_off=sel_off.astype(np.int32), table_sdf_sum=table0_sums[0], table_color_sum=table0_sums[1], **log.arrays())
    if snaps:
        out.update(snaps, kf_pool_idx=torch.stack(pool_idx).to(torch.int32))
    print(f"ref loop: ATE {100 * float(err.pow(2).mean().sqrt()):.2f} cm, max {100 * float(err.max()):.2f} cm at frame {int(err.argmax())}, "
          f"{len(mp.keyframe_list)} keyframes {mp.keyframe_list}, LC {int(mp.LC_cnt[0])}, tracking back at {[i for i, f in enumerate(tb_flags) if f]}, "
          f"{len(log.kind)} draws")
    return out


def g15():
    """the reference's loop (_ref_loop) with oracle/g15_settings.py G15: every frame tracked, mapped and kept as a keyframe"""
    npz("g15_sequence", **_ref_loop(G15))


def g16():
    """the reference's loop with G16: a mapped frame every 3rd, a keyframe every 2nd of those, ACTIVATED MAPPING on (uncertainty-triggered
    iteration doubling + tracking back: Tracker.py:352-363, Mapper.py:253-272,516) and a mapping window small enough for keyframe_selection_LC
    to choose while tracking back; + SNAPSHOTS of the loop's whole state at the start of a few frames, so that a replay can start from the
    reference's own state: one frame of the loop at a time, free of the Adam-amplified divergence a whole-sequence replay accumulates"""
    npz("g16_policy", **_ref_loop(G16))


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14, g15, g16):
        if not only or fn.__name__ in only:
            fn()
