"""
From the reference's YAML configuration to the objects the hot path runs on (reference src/config.py:21-71,
src/UNISLAM.py:63-88,168-259): config inheritance, intrinsics after the pre-processing crop, the enlarged scene bound, the grid
resolutions and per-level scale, the two hash-grid encoders, the decoders -- and `build_slam`, which hands them to the thin
Tracker / Mapper drivers of unislam_amd.slam together with a sequence reader.

The scene arithmetic (load_bound / get_resolution / per_level_scale / update_cam) is pinned by tests/golden/g13_scene.npz,
produced by the reference's own methods for its Replica room0, ScanNet scene0000 and TUM fr1_desk settings.
"""
import copy
import types

import numpy as np
import torch
import yaml

from .hashgrid import HashGridEncoding


# ---- src/config.py ------------------------------------------------------------------------------------------------
def update_recursive(base, override):
    """config.py:56-70: leaves of `override` replace those of `base`; nested dictionaries are merged level by level (in place)"""
    for key, value in override.items():
        if isinstance(value, dict):
            node = base.setdefault(key, {})
            if not isinstance(node, dict):              # a scalar in the parent gives way to a section in the child
                node = base[key] = {}
            update_recursive(node, value)
        else:
            base[key] = value


def load_config(path, default_path=None):
    """config.py:21-53: a file may name a parent (`inherit_from`), which is loaded first (recursively); a file without a parent starts
    from `default_path`, if one is given"""
    with open(path, "r") as f:
        own = yaml.full_load(f) or {}
    parent = own.get("inherit_from")
    if parent is not None:
        cfg = load_config(parent, default_path)
    elif default_path is not None:
        with open(default_path, "r") as f:
            cfg = yaml.full_load(f) or {}
    else:
        cfg = {}
    update_recursive(cfg, own)
    return cfg


# ---- src/UNISLAM.py:63-65,168-218 ---------------------------------------------------------------------------------
def update_cam(cfg):
    """(H, W, fx, fy, cx, cy) after `crop_size` (a resize: intrinsics scale) and `crop_edge` (UNISLAM.py:168-190)"""
    cam = cfg["cam"]
    H, W, fx, fy, cx, cy = cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    if "crop_size" in cam:
        crop = cam["crop_size"]
        sx, sy = crop[1] / W, crop[0] / H
        fx, fy, cx, cy = sx * fx, sy * fy, sx * cx, sy * cy
        W, H = crop[1], crop[0]
    e = cam["crop_edge"]
    if e > 0:
        H, W, cx, cy = H - 2 * e, W - 2 * e, cx - e, cy - e
    return H, W, fx, fy, cx, cy


def load_bound(cfg, scale=None):
    """UNISLAM.py:203-218: the configured bound times the global scale, upper corner moved up to a multiple of bound_dividable"""
    scale = cfg["scale"] if scale is None else scale
    bound = torch.from_numpy(np.array(cfg["mapping"]["bound"]) * scale).float()
    div = cfg["planes_res"]["bound_dividable"]
    bound[:, 1] = (((bound[:, 1] - bound[:, 0]) / div).int() + 1) * div + bound[:, 0]
    return bound


def grid_resolutions(cfg, bound):
    """UNISLAM.py:192-201: finest resolution = longest side / voxel size, truncated -> (sdf, colour)"""
    dim_max = (bound[:, 1] - bound[:, 0]).max()
    return int(dim_max / cfg["grid"]["voxel_sdf"]), int(dim_max / cfg["grid"]["voxel_color"])


def per_level_scale(desired_resolution, n_levels=16):
    """UNISLAM.py:241 -- as written there: log2(desired / n_levels), which equals log2(desired / base) for the shipped 16 / 16"""
    return np.exp2(np.log2(desired_resolution / n_levels) / (n_levels - 1))


def get_encoder(encoding_method="HashGrid", input_dim=3, n_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                desired_resolution=512):
    """UNISLAM.py:220-259 -> (encoder, n_output_dims)"""
    if not ("hash" in encoding_method.lower() or "tiled" in encoding_method.lower()):
        raise ValueError(f"get_encoder: only the hash grid exists in the reference ({encoding_method!r})")
    enc = HashGridEncoding(n_input_dims=input_dim, encoding_config={
        "otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": level_dim, "log2_hashmap_size": log2_hashmap_size,
        "base_resolution": base_resolution, "per_level_scale": per_level_scale(desired_resolution, n_levels)}, dtype=torch.float)
    return enc, enc.n_output_dims


def build_scene(cfg, device=None):
    """
    UNISLAM.__init__ (:63-88,127-137) without the process plumbing: intrinsics, decoders, bound, the two encoders on the device.
    Returns a namespace with H, W, fx, fy, cx, cy, scale, bound, resolution_sdf / _color, shared_decoders,
    shared_hash_grids_xyz / shared_c_hash_grids_xyz (1-element lists, as the reference keeps them), device.
    """
    from .decoders import get_model
    s = types.SimpleNamespace(cfg=cfg, device=device or cfg.get("device", "cuda:0"), scale=cfg["scale"])
    s.H, s.W, s.fx, s.fy, s.cx, s.cy = update_cam(cfg)
    s.shared_decoders = get_model(cfg)
    s.bound = load_bound(cfg, s.scale)
    s.shared_decoders.bound = s.bound
    if cfg["grid_mode"] != "hash_grid":
        raise ValueError("only grid_mode 'hash_grid' is functional in the reference (decoders.py:116-120)")
    s.resolution_sdf, s.resolution_color = grid_resolutions(cfg, s.bound)
    g = cfg["grid"]
    es, _ = get_encoder(g["enc"], log2_hashmap_size=g["hash_size_sdf"], desired_resolution=s.resolution_sdf)
    ec, _ = get_encoder(g["enc"], log2_hashmap_size=g["hash_size_color"], desired_resolution=s.resolution_color)
    s.shared_hash_grids_xyz, s.shared_c_hash_grids_xyz = [es.to(s.device)], [ec.to(s.device)]
    s.shared_decoders = s.shared_decoders.to(s.device)
    return s


def slam_options(cfg):
    """the reference's YAML keys -> the option dictionary of unislam_amd.slam.SLAM (same values, grouped names)"""
    t, m, r = cfg["tracking"], cfg["mapping"], cfg["rendering"]
    w = lambda c: dict(fs=c["w_sdf_fs"], center=c["w_sdf_center"], tail=c["w_sdf_tail"], depth=c["w_depth"], color=c["w_color"])
    return {
        "tracking": dict(ignore_edge_W=t["ignore_edge_W"], ignore_edge_H=t["ignore_edge_H"], const_speed_assumption=t["const_speed_assumption"],
                         lr_T=t["lr_T"], lr_R=t["lr_R"], pixels=t["pixels"], iters=t["iters"],
                         activated_mapping_mode=t.get("activated_mapping_mode", False), uncertainty_ts=t.get("uncertainty_ts", 0.001), w=w(t)),
        "mapping": dict(every_frame=m["every_frame"], keyframe_every=m["keyframe_every"], joint_opt=m["joint_opt"],
                        joint_opt_cam_lr=m["joint_opt_cam_lr"], mapping_window_size=m["mapping_window_size"],
                        lr_first_factor=m["lr_first_factor"], lr_factor=m["lr_factor"], pixels=m["pixels"], iters_first=m["iters_first"],
                        iters=m["iters"], LC=m.get("LC", False), LC_ts=m.get("LC_ts", 0.95),
                        lr=dict(decoders=m["lr"]["decoders_lr"], sdf_grid=m["lr"]["hash_grids_lr"], color_grid=m["lr"]["c_hash_grids_lr"]),
                        w=w(m)),
        "rendering": dict(n_stratified=r["n_stratified"], n_importance=r["n_importance"], perturb=r["perturb"]),
        "truncation": cfg["model"]["truncation"], "m_mask_mode": cfg["m_mask_mode"], "t_mask_mode": cfg["t_mask_mode"],
    }


def build_slam(cfg, args=None, frames=None, device=None):
    """the configured system: scene objects + sequence reader (`frames` overrides the reader, e.g. a synthetic source) -> slam.SLAM"""
    from .datasets import get_dataset
    from .slam import SLAM
    cfg = copy.deepcopy(cfg)
    s = build_scene(cfg, device)
    if frames is None:
        frames = get_dataset(cfg, args, s.scale, device=s.device)
    return SLAM(frames, (s.H, s.W, s.fx, s.fy, s.cx, s.cy), s.shared_hash_grids_xyz[0], s.shared_c_hash_grids_xyz[0], s.shared_decoders,
                s.bound, cfg=slam_options(cfg))
