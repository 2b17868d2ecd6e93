"""
unislam_amd -- MI355X (gfx950) implementation of Uni-SLAM's per-iteration volumetric-rendering hot path.

Python host side (this package) mirrors the reference's module/function API for the path; all compute runs in
hand-written HIP kernels behind the C ABI of include/unislam_hip.h (libunislam_hip.so, loaded with ctypes).
Importing the package does not need a GPU; calling an op without the built library or with CPU tensors raises.
"""
from . import _lib
from ._lib import UniSlamHipError, LIB_PATH
from .hashgrid import HashGridEncoding, make_grid_desc, grid_indices
from .network import FusedMLP, fused_mlp, make_mlp_desc
from .decoders import Decoders, get_model
from .renderer import Renderer, sample_z
from .losses import sdf_losses, mapping_loss, tracking_loss, fused_loss
from .mapstep import MapStep
from .trackstep import TrackStep
from .window import MapWindow, ArenaWindow, KeyframeArena
from .graph import CapturedIteration
from .mesher import eval_points
from . import common, tcnn, optim

__all__ = ["HashGridEncoding", "FusedMLP", "Decoders", "Renderer", "sdf_losses", "mapping_loss", "tracking_loss",
           "fused_loss", "common", "tcnn", "UniSlamHipError", "LIB_PATH", "get_model", "make_grid_desc",
           "grid_indices", "fused_mlp", "make_mlp_desc", "sample_z", "MapStep", "MapWindow", "ArenaWindow", "KeyframeArena", "TrackStep", "CapturedIteration", "eval_points"]
