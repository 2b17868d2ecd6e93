"""unislam_amd.config: scene arithmetic against the fixture produced by the reference's UNISLAM methods on its own configs (g13),
YAML inheritance, option translation.  CPU only."""
import numpy as np
import pytest
import torch
import yaml

from unislam_amd import config as C

TAGS = ("room0", "scene0000", "fr1_desk")


@pytest.mark.parametrize("tag", TAGS)
def test_g13_scene_arithmetic(golden, tag):
    g = golden("g13_scene")
    H, W, fx, fy, cx, cy, edge, c0, c1 = g[f"{tag}_cam_in"]
    cam = dict(H=int(H), W=int(W), fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), crop_edge=int(edge))
    if c0 > 0:
        cam["crop_size"] = [int(c0), int(c1)]
    v_sdf, v_col, div, scale = g[f"{tag}_voxel"]
    cfg = {"cam": cam, "scale": float(scale) if scale != int(scale) else int(scale), "mapping": {"bound": g[f"{tag}_bound_in"].tolist()},
           "planes_res": {"bound_dividable": float(div)}, "grid": {"voxel_sdf": float(v_sdf), "voxel_color": float(v_col)}}
    np.testing.assert_array_equal(np.array(C.update_cam(cfg), dtype=np.float64), g[f"{tag}_cam"])
    bound = C.load_bound(cfg)
    np.testing.assert_array_equal(bound.numpy(), g[f"{tag}_bound"])
    res = C.grid_resolutions(cfg, bound)
    np.testing.assert_array_equal(np.array(res), g[f"{tag}_res"])
    np.testing.assert_array_equal(np.array([C.per_level_scale(r) for r in res]), g[f"{tag}_pls"])


def _write(path, obj):
    path.write_text(yaml.safe_dump(obj))
    return str(path)


def test_load_config_inheritance_and_options(tmp_path):
    base = {"scale": 1, "m_mask_mode": "original", "t_mask_mode": "original", "grid_mode": "hash_grid",
            "planes_res": {"bound_dividable": 0.24},
            "grid": {"enc": "HashGrid", "hash_size_sdf": 19, "hash_size_color": 19, "voxel_sdf": 0.01, "voxel_color": 0.01, "tcnn_network": False},
            "tracking": {"ignore_edge_W": 75, "ignore_edge_H": 75, "const_speed_assumption": True, "lr_T": 0.001, "lr_R": 0.001, "pixels": 2000,
                         "iters": 8, "w_sdf_fs": 10, "w_sdf_center": 200, "w_sdf_tail": 50, "w_depth": 1, "w_color": 5},
            "mapping": {"every_frame": 4, "joint_opt": True, "joint_opt_cam_lr": 0.001, "keyframe_every": 4, "mapping_window_size": 20,
                        "lr_first_factor": 5, "lr_factor": 1, "pixels": 4000, "iters_first": 10, "iters": 15, "w_sdf_fs": 5, "w_sdf_center": 200,
                        "w_sdf_tail": 10, "w_depth": 0.1, "w_color": 5, "LC": True,
                        "lr": {"decoders_lr": 0.001, "hash_grids_lr": 0.05, "c_hash_grids_lr": 0.05}},
            "cam": {"H": 680, "W": 1200, "fx": 600.0, "fy": 600.0, "cx": 599.5, "cy": 339.5, "png_depth_scale": 6553.5, "crop_edge": 0},
            "rendering": {"n_stratified": 32, "n_importance": 8, "perturb": True, "learnable_beta": True},
            "model": {"c_dim": 32, "truncation": 0.06}}
    default = _write(tmp_path / "default.yaml", base)
    family = _write(tmp_path / "family.yaml", {"dataset": "replica", "tracking": {"lr_T": 0.002, "activated_mapping_mode": True, "uncertainty_ts": 0.001},
                                               "mapping": {"lr": {"hash_grids_lr": 0.02}, "LC_ts": 0.9}, "grid": {"hash_size_sdf": 16}})
    scene = _write(tmp_path / "scene.yaml", {"inherit_from": family, "mapping": {"bound": [[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]]},
                                             "data": {"input_folder": "x"}})
    cfg = C.load_config(scene, default)
    assert cfg["tracking"]["lr_T"] == 0.002 and cfg["tracking"]["lr_R"] == 0.001 and cfg["tracking"]["pixels"] == 2000
    assert cfg["mapping"]["lr"] == {"decoders_lr": 0.001, "hash_grids_lr": 0.02, "c_hash_grids_lr": 0.05}
    assert cfg["grid"]["hash_size_sdf"] == 16 and cfg["grid"]["hash_size_color"] == 19 and cfg["dataset"] == "replica"
    assert cfg["inherit_from"] == family and cfg["mapping"]["bound"][0] == [-1.0, 7.0]
    assert C.load_config(family) == yaml.safe_load(open(family))                  # no parent, no default: the file itself
    o = C.slam_options(cfg)
    assert o["tracking"]["w"] == dict(fs=10, center=200, tail=50, depth=1, color=5) and o["tracking"]["lr_T"] == 0.002
    assert o["tracking"]["activated_mapping_mode"] is True and o["mapping"]["LC_ts"] == 0.9
    assert o["mapping"]["lr"] == dict(decoders=0.001, sdf_grid=0.02, color_grid=0.05) and o["mapping"]["w"]["depth"] == 0.1
    assert o["truncation"] == 0.06 and o["rendering"] == dict(n_stratified=32, n_importance=8, perturb=True)
    # the slam driver's defaults are the reference's Replica settings: the translation of this file reproduces their key set
    from unislam_amd.slam import DEFAULTS
    for k in ("tracking", "mapping", "rendering"):
        assert set(o[k]) == set(DEFAULTS[k]), (k, set(o[k]) ^ set(DEFAULTS[k]))
    # encoders: level tables of the room0 configuration
    bound = C.load_bound(cfg)
    res_s, res_c = C.grid_resolutions(cfg, bound)
    assert (res_s, res_c) == (816, 816)
    enc, n_out = C.get_encoder("HashGrid", log2_hashmap_size=16, desired_resolution=res_s)
    assert n_out == 32 and enc.params.numel() == 1736800
    with pytest.raises(ValueError):
        C.get_encoder("freq")
