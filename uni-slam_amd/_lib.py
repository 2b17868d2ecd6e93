"""
ctypes binding of libunislam_hip.so (C ABI: include/unislam_hip.h).

There is NO CPU fallback: if the shared library is missing, or a tensor is not on a HIP device, the calls raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("US_LIB_PATH") or os.path.join(_HERE, "libunislam_hip.so")     # (US_LIB_PATH: a timing / experiments build, development only)
US_MAX_LEVELS = 32
US_ERR_CONFIG = -3          # unislam_hip.h: unsupported descriptor / configuration
US_GRID_CLAMP01 = 1
US_GRID_LEVEL_MAJOR = 2
US_GRID_FEAT_SPLIT_BF16 = 2048  # us_hashgrid_fwd_joint(_dydx): feature planes as hi / lo bf16 pairs (the split-bf16 decoders' operands)
US_GRID_BWD_OVERWRITE = 4
US_GRID_ACCUMULATE = 8
US_GRID_BWD_COUNTED = 16
US_GRID_BWD_PACKED = 32
US_GRID_BWD_SCANNED = 64
US_GRID_BWD_DETERMINISTIC = 128
US_GRID_BWD_ONLY_A = 256
US_GRID_BWD_ONLY_B = 512
US_GRID_BWD_RECORDS_READY = 1024
US_MLP_LEVEL_MAJOR = 1
US_MLP_DEFER_REDUCE = 2
FEAT_SPLIT_DEFAULT = os.environ.get("US_FEAT_SPLIT", "1") != "0"   # the pre-split feature hand-over of the split-bf16 decoders (A/B switch)
US_MLP_OUT_PREACT, US_MLP_DOUT_PREACT = 8, 16   # decoder outputs before out_act / dL_dout w.r.t. them (the compositing launches take the activation)
US_RENDER_ACT_ON = 0x100000
ACT_HANDOVER_DEFAULT = os.environ.get("US_ACT_HANDOVER", "1") != "0"    # A/B switch of that hand-over
US_MLP_IN_SPLIT_BF16 = 4        # `in` holds the hi / lo bf16 pairs the joint encoder wrote with US_GRID_FEAT_SPLIT_BF16
US_LOSS_DEFER_BETA = 256
US_ADAM_STEP_ADVANCED = 0x80000000
US_POSE_GRAD_ONLY = 1
US_POSE_OWN_STEP = 2

c_f = ctypes.c_void_p          # device pointers travel as void*
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_u32 = ctypes.c_uint32
c_flt = ctypes.c_float
c_dbl = ctypes.c_double


class GridDesc(ctypes.Structure):
    """us_grid_desc"""
    _fields_ = [
        ("n_levels", c_u32), ("n_features", c_u32), ("log2_hashmap_size", c_u32), ("base_resolution", c_u32),
        ("per_level_scale", c_flt), ("scale", c_flt * US_MAX_LEVELS), ("resolution", c_u32 * US_MAX_LEVELS),
        ("offset", c_u32 * (US_MAX_LEVELS + 1)), ("n_params", c_u32),
    ]


class MlpDesc(ctypes.Structure):
    """us_mlp_desc"""
    _fields_ = [("n_in", c_u32), ("width", c_u32), ("n_hidden", c_u32), ("n_out", c_u32), ("out_act", c_u32),
                ("has_bias", c_u32), ("precision", c_u32)]


_GP = ctypes.POINTER(GridDesc)
_MP = ctypes.POINTER(MlpDesc)

class PoseStepDesc(ctypes.Structure):
    """include/unislam_hip.h: us_pose_step_desc -- a joint_opt window's pose group riding in us_adam_step_model's launch"""
    _fields_ = [("poses7", ctypes.c_void_p), ("n_poses", ctypes.c_int), ("g_rays_o", ctypes.c_void_p), ("g_rays_d", ctypes.c_void_p),
                ("dirs", ctypes.c_void_p), ("row_a", ctypes.c_int64), ("n_a", ctypes.c_int64), ("first_pose_b", ctypes.c_int),
                ("row_b", ctypes.c_int64), ("n_b", ctypes.c_int64), ("m7", ctypes.c_void_p), ("v7", ctypes.c_void_p), ("g7_out", ctypes.c_void_p),
                ("lr_q", ctypes.c_double), ("lr_t", ctypes.c_double), ("shape_dev", ctypes.c_void_p), ("rows_a", ctypes.c_int64)]


class TableAdamDesc(ctypes.Structure):
    """include/unislam_hip_experiments.h: us_table_adam_desc -- the tables' Adam step applied inside the accumulate pass (us_hashgrid_bwd_joint_adam)"""
    _fields_ = [("pA", ctypes.c_void_p), ("mA", ctypes.c_void_p), ("vA", ctypes.c_void_p), ("pB", ctypes.c_void_p), ("mB", ctypes.c_void_p),
                ("vB", ctypes.c_void_p), ("lrA", ctypes.c_double), ("lrB", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("eps", ctypes.c_double), ("step_dev", ctypes.c_void_p), ("write_grad", ctypes.c_int)]


_HF = ctypes.POINTER(c_flt)     # host float array

# name -> (restype, argtypes): exactly the declarations of include/unislam_hip.h
SIGNATURES = {
    "us_last_error": (ctypes.c_char_p, []),
    "us_abi_version": (c_int, []),
    "us_grid_desc_init": (c_int, [_GP, c_u32, c_u32, c_u32, c_u32, c_flt]),
    "us_hashgrid_fwd": (c_int, [_GP, c_f, c_f, c_i64, c_f, c_f, c_int, c_f]),
    "us_hashgrid_indices": (c_int, [_GP, c_f, c_i64, c_f, c_int, c_f]),
    "us_hashgrid_bwd_params": (c_int, [_GP, c_f, c_f, c_i64, c_f, c_int, c_int, c_f]),
    "us_hashgrid_bwd_workspace_bytes": (ctypes.c_size_t, [_GP, c_i64]),
    "us_hashgrid_bwd_binned_supported": (c_int, [_GP, c_i64]),
    "us_hashgrid_bwd_binned": (c_int, [_GP, c_f, c_f, c_i64, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_fwd_counted": (c_int, [_GP, c_f, c_f, c_i64, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_bwd_scan": (c_int, [_GP, c_i64, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_joint_supported": (c_int, [_GP, _GP, c_i64]),
    "us_hashgrid_joint_workspace_bytes": (ctypes.c_size_t, [_GP, _GP, c_i64]),
    "us_hashgrid_fwd_joint": (c_int, [_GP, _GP, c_f, c_f, c_f, c_i64, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_bwd_joint": (c_int, [_GP, _GP, c_f, c_f, c_f, c_i64, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_bwd_joint_img": (c_int, [_GP, _GP, c_f, c_f, c_f, c_i64, c_f, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_bwd_joint_range": (c_int, [_GP, _GP, c_f, c_f, c_f, c_i64, c_i64, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_bwd_binned_range": (c_int, [_GP, c_f, c_f, c_i64, c_i64, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_joint_scan": (c_int, [_GP, _GP, c_i64, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_bwd_input": (c_int, [c_f, c_f, c_i64, c_u32, c_f, c_f]),
    "us_hashgrid_bwd_input_gather": (c_int, [_GP, c_f, c_f, c_f, c_i64, c_f, c_int, c_f]),
    "us_mlp_n_params": (ctypes.c_size_t, [_MP]),
    "us_mlp_fwd": (c_int, [_MP, c_f, c_f, c_i64, c_f, c_i64, c_int, c_f]),
    "us_mlp_pair_supported": (c_int, [_MP, _MP]),
    "us_mlp_reduce_pair": (c_int, [_MP, _MP, c_f, c_f, ctypes.c_size_t, c_i64, c_f, c_f, c_f]),
    "us_mlp_bwd_pair_dydx": (c_int, [_MP, _MP, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_i64, c_f, c_i64, c_f, c_i64, c_i64, c_f, c_f, c_f, c_f, c_int,
                                     c_f, c_f, ctypes.c_size_t, c_f, c_f, c_f, c_f, c_f]),
    "us_ray_points_bwd2": (c_int, [c_f, c_f, c_f, _HF, c_i64, c_int, c_f, c_f, c_f]),
    "us_mlp_reduce_pair_adam": (c_int, [_MP, _MP, c_f, c_f, ctypes.c_size_t, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_f, c_f,
                                        c_dbl, c_dbl, c_dbl, c_dbl, c_f, c_f]),
    "us_adam_step_model": (c_int, [_MP, _MP, c_f, c_f, ctypes.c_size_t, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_f, c_f,
                                   c_dbl, c_f, c_f, c_f, c_f, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl),
                                   c_dbl, c_dbl, c_dbl, c_f, ctypes.c_uint, ctypes.POINTER(PoseStepDesc), c_f]),
    "us_mlp_fwd_pair": (c_int, [_MP, _MP, c_f, c_f, c_f, c_f, c_i64, c_f, c_i64, c_f, c_i64, c_int, c_f]),
    "us_mlp_bwd_pair": (c_int, [_MP, _MP, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_i64, c_f, c_i64, c_f, c_i64, c_i64, c_f, c_f, c_f, c_f, c_int,
                                c_f, c_f, ctypes.c_size_t, c_f]),
    "us_mlp_bwd_workspace_bytes": (ctypes.c_size_t, [_MP]),
    "us_mlp_reduce": (c_int, [_MP, c_f, ctypes.c_size_t, c_i64, c_f, c_f]),
    "us_beta_reduce": (c_int, [c_f, c_i64, c_f, c_f]),
    "us_adam_step_inc": (c_int, [c_f, c_dbl, c_dbl, c_f]),
    "us_mlp_bwd": (c_int, [_MP, c_f, c_f, c_f, c_i64, c_f, c_i64, c_i64, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_sample_z": (c_int, [c_f, c_i64, c_f, c_int, c_f, c_int, c_flt, c_flt, c_flt, c_f, c_f, c_f]),
    "us_ray_points": (c_int, [c_f, c_f, c_f, _HF, c_i64, c_int, c_f, c_f]),
    "us_ray_points_bwd": (c_int, [c_f, c_f, _HF, c_i64, c_int, c_f, c_f, c_f]),
    "us_bbox_filter": (c_int, [c_f, c_f, c_f, _HF, c_i64, c_int, c_f, c_f, c_f]),
    "us_gather_rays": (c_int, [c_f, c_f, c_f, c_f, c_f, c_int, c_i64, c_i64, c_f, c_f, c_f, c_f, c_f]),
    "us_composite_fwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_composite_bwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_loss_partials_size": (ctypes.c_size_t, [c_i64]),
    "us_loss_stats": (c_int, [c_int, c_f, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_dbl, c_f, c_f, c_f]),
    "us_loss_grad": (c_int, [c_int, c_f, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_dbl, _HF, c_f,
                             c_f, c_f, c_f, c_f, c_f]),
    "us_render_loss_fwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_f, c_f, c_f, c_dbl, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_render_loss_bwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_dbl, _HF, c_f, c_f, c_f, c_f, c_f,
                                   c_f]),
    "us_pose_adam_step": (c_int, [c_f, c_f, c_f, c_f, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_f, c_f]),
    "us_masked_median": (c_int, [c_f, c_f, c_f, c_i64, c_f, c_f]),
    "us_masked_mean": (c_int, [c_f, c_f, c_i64, c_f, c_f]),
    "us_adam_step": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_f]),
    "us_adam_step_tensors": (c_int, [c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl), c_dbl, c_dbl, c_dbl, c_int, c_f]),
    "us_adam_step_segments": (c_int, [c_f, c_f, c_f, c_f, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                      ctypes.POINTER(c_dbl), c_dbl, c_dbl, c_dbl, c_int, ctypes.c_uint, c_f]),
    "us_adam_step_segments_dev": (c_int, [c_f, c_f, c_f, c_f, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                          ctypes.POINTER(c_dbl), c_dbl, c_dbl, c_dbl, c_f, ctypes.c_uint, c_f]),
    "us_adam_step_segments_bf16": (c_int, [c_f, c_f, c_f, ctypes.c_uint, c_f, c_f, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                           ctypes.POINTER(c_dbl), c_dbl, c_dbl, c_dbl, c_f, ctypes.c_uint, c_f]),
    "us_importance_z": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_f, c_f]),
    "us_zero_depth_rows": (c_int, [c_f, c_i64, c_f, c_f, c_f]),
    "us_uniform_points": (c_int, [c_f, c_f, c_f, c_i64, _HF, c_f, c_int, c_f, ctypes.c_uint64, c_int, c_f, c_f, c_f]),
    "us_importance_z_rows": (c_int, [c_f, c_f, c_f, c_f, ctypes.c_uint64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, _HF, c_f, c_f]),
    "us_zero_depth_resample": (c_int, [_GP, c_f, _MP, c_f, c_f, c_f, c_f, c_f, c_i64, _HF, c_f, c_int, c_int, c_f, c_f, ctypes.c_uint64,
                                       ctypes.c_uint64, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_sample_points": (c_int, [c_f, c_f, c_f, _HF, c_i64, c_f, c_int, c_f, c_int, c_flt, c_flt, c_flt, c_f, ctypes.c_uint64,
                                 c_f, c_int, c_int, c_f, c_f, c_f, c_f]),
    "us_adam_step_dev": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_dbl, c_dbl, c_dbl, c_dbl, c_f, c_f]),
    "us_pose_rays": (c_int, [c_f, c_f, c_i64, _HF, c_int, c_int, c_int, c_f, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_pose_grad": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_f, c_f]),
    "us_hashgrid_bwd_input_rays_supported": (c_int, [_GP, _GP, c_int]),
    "us_hashgrid_bwd_input_rays": (c_int, [_GP, _GP, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_f, _HF, c_f, c_f, c_f, c_int, c_f]),
    "us_hashgrid_fwd_joint_dydx": (c_int, [_GP, _GP, c_f, c_f, c_f, c_i64, c_f, c_f, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_dydx_rays": (c_int, [c_u32, c_f, c_f, c_f, c_f, c_i64, c_int, c_f, _HF, c_f, c_f, c_f, c_f]),
    "us_track_sample": (c_int, [c_f, c_f, c_i64, _HF, c_int, c_int, c_int, c_int, c_f, c_f, c_int, _HF, c_f, c_int, c_f, c_int, c_flt, c_flt, c_flt,
                                c_f, ctypes.c_uint64, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_track_loss_fwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_f, c_dbl, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_track_loss_bwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_dbl, _HF, c_f, c_f, c_f, c_f]),
    "us_window_sample": (c_int, [c_f, c_f, c_int, c_i64, c_int, c_i64, c_f, c_f, c_f, c_i64, c_f, c_f, _HF, c_f, c_int, c_f, c_int, c_flt, c_flt, c_flt,
                                 c_f, ctypes.c_uint64, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_arena_window_sample": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_f, c_f, c_f, c_i64, c_f, c_f, _HF, c_f, c_int, c_f, c_int, c_flt, c_flt, c_flt,
                                       c_f, ctypes.c_uint64, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_arena_pose_step": (c_int, [c_f, c_int, c_f, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_f, c_f]),
    "us_pool_cut": (c_int, [c_f, c_f, c_f, c_i64, c_i64, ctypes.c_uint64, c_f, c_f, c_f, c_f, c_f]),
    "us_keyframe_overlap": (c_int, [c_f, c_f, c_f, c_int, c_int, _HF, c_int, c_int, c_int, c_f, c_f, c_int, c_f, c_f]),
    "us_window_rays": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_i64, c_f, c_f, c_f, c_f, c_f, c_f]),
    "us_pose_window_step": (c_int, [c_f, c_int, c_f, c_f, c_f, c_i64, c_i64, c_int, c_i64, c_i64, c_f, c_f, c_f, c_dbl, c_dbl, c_dbl, c_dbl,
                                    c_dbl, c_f, c_int, c_f]),
    "us_matrix_to_cam_pose": (c_int, [c_f, c_int, c_int, c_f, c_f]),
    "us_cam_pose_to_matrix": (c_int, [c_f, c_int, c_f, c_f]),
    "us_pose_track_step": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_f, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_f, c_f, c_f, c_f, c_f, c_f]),
}

# the experiments build (tools/build_experiments.sh, include/unislam_hip_experiments.h): bound when the loaded library exports them
EXPERIMENT_SIGNATURES = {
    "us_encode_decode_supported": (c_int, [_GP, _GP, _MP, _MP]),
    "us_hashgrid_bwd_joint_adam": (c_int, [_GP, _GP, c_f, c_f, c_f, c_i64, c_f, c_f, ctypes.POINTER(TableAdamDesc), c_int, c_f, ctypes.c_size_t, c_f]),
    "us_hashgrid_bwd_joint_part": (c_int, [_GP, _GP, c_f, c_f, c_f, c_i64, c_f, c_f, c_int, c_f, ctypes.c_size_t, c_int, c_int, c_int, c_f]),
    "us_encode_decode_fwd": (c_int, [_GP, _GP, c_f, c_f, _MP, _MP, c_f, c_f, c_f, c_i64, c_f, c_i64, c_f, c_i64, c_int, c_f]),
}

_lib = None


class UniSlamHipError(RuntimeError):
    pass


def lib():
    """Load libunislam_hip.so (built by __graft_entry__.build() / make -C uni-slam_amd/csrc).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UniSlamHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  unislam_amd has no CPU or PyTorch fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in EXPERIMENT_SIGNATURES.items():
            fn = getattr(l, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = l
    return _lib


def has_experiments():
    """True when the loaded library is the experiments build (exports the measured-slower variants)"""
    return hasattr(lib(), "us_encode_decode_fwd")


def check(rc, what=""):
    if rc != 0:
        msg = lib().us_last_error()
        raise UniSlamHipError(f"{what}: rc={rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a contiguous fp32/int64/uint8 HIP tensor (None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise UniSlamHipError("unislam_amd kernels run on the GPU only: got a tensor on " + str(t.device))
    if not t.is_contiguous():
        raise UniSlamHipError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def f32(t):
    """contiguous fp32 view/copy on the same device"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def host_floats(vals):
    arr = (c_flt * len(vals))(*[float(v) for v in vals])
    return arr
