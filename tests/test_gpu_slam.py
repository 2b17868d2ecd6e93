"""
GPU: the thin tracking / mapping drivers (unislam_amd.slam, after src/Tracker.py:271-370 and src/Mapper.py:177-545) on a synthetic
RGB-D sequence (unislam_amd.synthetic): the trajectory is recovered, tracking beats dead reckoning, the keyframe machinery and
the joint pose optimisation run.
"""
import numpy as np
import pytest
import torch

import unislam_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(us, n_frames, seed=0, mlp_precision="fp32"):
    from unislam_amd.synthetic import SyntheticRoom
    from unislam_amd.slam import SLAM
    torch.manual_seed(seed)
    frames = SyntheticRoom(n_frames=n_frames, H=120, W=160, device=DEV)
    bound = O.load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]])
    res = int((bound[:, 1] - bound[:, 0]).max() / 0.02)
    ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                       "base_resolution": 16, "per_level_scale": O.per_level_scale(res)}
    es, ec = us.HashGridEncoding(3, ecfg(16)).to(DEV), us.HashGridEncoding(3, ecfg(16)).to(DEV)
    cfg = {"rendering": {"perturb": True, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid",
           "grid": {"tcnn_network": False}, "model": {"mlp_precision": mlp_precision}}
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06).to(DEV)
    dec.bound = bound
    slam = SLAM(frames, (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy), es, ec, dec, bound,
                cfg={"tracking": dict(ignore_edge_W=8, ignore_edge_H=8, pixels=1000, iters=10),
                     "mapping": dict(pixels=2000, iters=20, iters_first=300, every_frame=2, keyframe_every=2)})
    return slam, frames


def test_slam_recovers_the_trajectory():
    import unislam_amd as us
    n = 26
    slam, frames = _build(us, n)
    est = slam.run()
    print('ATE', slam.ate_rmse(), 'LC', slam.mapper.LC_cnt, 'kf', slam.mapper.keyframe_list)
    gt = slam.gt_c2w_list[:n]
    ate = slam.ate_rmse()
    # dead reckoning from frame 1 on (constant velocity with the first two true poses) drifts much further
    travelled = float((gt[1:, :3, 3] - gt[:-1, :3, 3]).norm(dim=-1).sum())
    assert travelled > 0.4
    assert ate < 0.02, ate                                                  # 2 cm over ~0.5 m of motion
    _, res = slam.evaluate()                                                # the reference's report: after Horn alignment, in cm
    print(res)
    assert res['compared_pose_pairs'] == n and res['error.rmse'] <= 100 * ate + 0.01
    rot_err = torch.linalg.matrix_norm(est[:, :3, :3] - gt[:, :3, :3]).max()
    assert float(rot_err) < 0.05
    m = slam.mapper
    assert len(m.keyframe_list) >= n // 2 - 1 and m.keyframe_list[0] == 0
    assert m.joint_opt                                                      # > 4 keyframes: window poses optimised jointly
    # the map explains the last frame: render it at the ESTIMATED pose and compare with the measured depth
    rend = us.Renderer({"rendering": {"perturb": False, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid"},
                       type("U", (), dict(bound=slam.bound, device=DEV, H=frames.H, W=frames.W, fx=frames.fx, fy=frames.fy,
                                          cx=frames.cx, cy=frames.cy))())
    _, color, depth, _, _ = frames[n - 1]
    with torch.no_grad():
        out = rend.render_img(([slam.es], [slam.ec]), slam.decoders, est[n - 1], 0.06, DEV, gt_depth=depth)
    d_hat = out[0]
    err = (d_hat.float().reshape(-1) - depth.reshape(-1)).abs()
    assert float(err.median()) < 0.03, float(err.median())


def test_keyframe_selection_on_device():
    import unislam_amd as us
    from unislam_amd.slam import keyframe_selection_LC
    from unislam_amd.synthetic import SyntheticRoom
    torch.manual_seed(1)
    frames = SyntheticRoom(n_frames=4, H=120, W=160, device=DEV)
    _, color, depth, c2w, _ = frames[0]
    cam = (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy)
    # keyframes: the same view, a slightly shifted one, one looking the opposite way (+ two trailing entries, always excluded)
    same, shifted, away = c2w.clone(), c2w.clone(), c2w.clone()
    shifted[:3, 3] += torch.tensor([0.05, 0.02, 0.0], device=DEV)
    away[:3, 0] *= -1; away[:3, 2] *= -1
    est = torch.stack([same, shifted, away, same, same])
    sel, pct, loop = keyframe_selection_LC(3, 10, color, depth, c2w, 2, [0, 1, 2, 3, 4], est, cam, DEV)
    # the same camera sees every sample at its own pixel: inside unless within the 20-pixel border, (120*80)/(160*120) = 0.5
    assert 0.35 < float(pct[0]) < 0.65 and float(pct[1]) > 0.25 and float(pct[2]) == 0.0 and not loop
    assert sel == [0, 1, 2]                                                 # "global": every keyframe joins the window
    sel, pct, loop = keyframe_selection_LC(3, 10, color, depth, c2w, 2, [0, 1, 2, 3, 4], est, cam, DEV, tracking_back=True)
    assert sorted(sel) == [0, 1]                                            # tracking back: the best-overlapping ones


def test_slam_from_a_sequence_on_disk(tmp_path):
    """the same loop fed by the Replica-layout reader (unislam_amd.datasets) from files written by export_sequence: JPEG colour,
    16-bit PNG depth, traj.txt poses"""
    import unislam_amd as us
    from unislam_amd import datasets as D
    from unislam_amd.slam import SLAM
    n = 12
    slam0, frames = _build(us, n)
    folder = D.export_sequence(frames, str(tmp_path / "room"), layout="replica", png_depth_scale=6553.5)
    cam = dict(H=frames.H, W=frames.W, fx=frames.fx, fy=frames.fy, cx=frames.cx, cy=frames.cy, png_depth_scale=6553.5, crop_edge=0)
    ds = D.get_dataset({"dataset": "replica", "cam": cam, "data": {"input_folder": folder}}, None, 1.0, device=DEV)
    assert len(ds) == n and ds[0][1].device.type == "cpu"
    H, W, fx, fy, cx, cy = ds.intrinsics()
    slam = SLAM(ds, (H, W, fx, fy, cx, cy), slam0.es, slam0.ec, slam0.decoders, slam0.bound, cfg=slam0.cfg)
    slam.run()
    ate = slam.ate_rmse()
    print("ATE from disk", ate)
    assert ate < 0.02, ate
