"""
oracle/unislam_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy / torch-CPU / the C file next to it) of Uni-SLAM's per-iteration
volumetric-rendering hot path.  Every function cites the reference file:line it follows
(paths relative to /root/reference).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product package never does.

Pinning status
  * everything that lives in the reference's own Python (rays, z-sampling, compositing,
    torch-MLP decoder, losses) is pinned against outputs of the reference itself, captured by
    oracle/gen_golden.py into tests/golden/*.npz (tests/test_oracle_golden.py);
  * the hash-grid encoding and the FullyFusedMLP live in the un-vendored tiny-cuda-nn
    (requirements.txt:90, commit 2ec562e8...).  Their restatement follows the published
    algorithm and is PARITY UNPINNED (no reference fixture exists, the dependency cannot
    run here); it is held by spec-derived known-answer tests only.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
ORC_MAX_LEVELS = 32


# --------------------------------------------------------------------------------------
# scene bound / grid resolution arithmetic (src/UNISLAM.py:192-222, :241)
# --------------------------------------------------------------------------------------
def load_bound(bound, scale=1.0, bound_dividable=0.24):
    """src/UNISLAM.py:205-218: enlarge the upper bound so every side divides by bound_dividable."""
    b = torch.from_numpy(np.array(bound, dtype=np.float64) * scale).float()
    b[:, 1] = (((b[:, 1] - b[:, 0]) / bound_dividable).int() + 1) * bound_dividable + b[:, 0]
    return b


def get_resolution(bound, voxel):
    """src/UNISLAM.py:192-199: int(longest side / voxel size)."""
    dim_max = (bound[:, 1] - bound[:, 0]).max()
    return int(dim_max / voxel)


def per_level_scale(desired_resolution, n_levels=16):
    """src/UNISLAM.py:241 (note: divides by n_levels, which equals base_resolution only by coincidence)."""
    return float(np.exp2(np.log2(desired_resolution / n_levels) / (n_levels - 1)))


# --------------------------------------------------------------------------------------
# hash grid: C restatement loader
# --------------------------------------------------------------------------------------
class GridDesc(ctypes.Structure):
    _fields_ = [
        ("n_levels", ctypes.c_uint32),
        ("n_features", ctypes.c_uint32),
        ("log2_hashmap_size", ctypes.c_uint32),
        ("base_resolution", ctypes.c_uint32),
        ("per_level_scale", ctypes.c_float),
        ("scale", ctypes.c_float * ORC_MAX_LEVELS),
        ("resolution", ctypes.c_uint32 * ORC_MAX_LEVELS),
        ("offset", ctypes.c_uint32 * (ORC_MAX_LEVELS + 1)),
        ("n_params", ctypes.c_uint32),
    ]


_clib = None


def build_clib(force=False):
    """Compile oracle/hashgrid_ref.c with gcc (recipe: oracle/Makefile)."""
    so = os.path.join(_HERE, "libhashgrid_ref.so")
    src = os.path.join(_HERE, "hashgrid_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libhashgrid_ref.so"])
    return so


def clib():
    global _clib
    if _clib is None:
        lib = ctypes.CDLL(build_clib())
        fp = ctypes.POINTER(ctypes.c_float)
        up = ctypes.POINTER(ctypes.c_uint32)
        dp = ctypes.POINTER(GridDesc)
        lib.orc_grid_desc_init.argtypes = [dp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float]
        lib.orc_grid_desc_init.restype = ctypes.c_int
        lib.orc_hashgrid_indices.argtypes = [dp, fp, ctypes.c_int64, up]
        lib.orc_hashgrid_fwd.argtypes = [dp, fp, fp, ctypes.c_int64, fp, fp]
        lib.orc_hashgrid_bwd_params.argtypes = [dp, fp, fp, ctypes.c_int64, fp]
        lib.orc_hashgrid_bwd_input.argtypes = [fp, fp, ctypes.c_int64, ctypes.c_uint32, fp]
        for f in (lib.orc_hashgrid_indices, lib.orc_hashgrid_fwd, lib.orc_hashgrid_bwd_params, lib.orc_hashgrid_bwd_input):
            f.restype = None
        _clib = lib
    return _clib


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def make_grid_desc(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=2.0):
    d = GridDesc()
    rc = clib().orc_grid_desc_init(ctypes.byref(d), n_levels, n_features, log2_hashmap_size, base_resolution,
                                   ctypes.c_float(per_level_scale))
    if rc != 0:
        raise ValueError("bad grid configuration")
    return d


def desc_tables(d):
    L = d.n_levels
    return (np.array(d.scale[:L], dtype=np.float32), np.array(d.resolution[:L], dtype=np.uint32),
            np.array(d.offset[:L + 1], dtype=np.uint32))


def hashgrid_indices(d, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((x.shape[0], d.n_levels, 8), dtype=np.uint32)
    clib().orc_hashgrid_indices(ctypes.byref(d), _fptr(x), x.shape[0], out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return out


def hashgrid_fwd(d, params, x, want_dydx=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    params = np.ascontiguousarray(params, dtype=np.float32)
    assert params.size == d.n_params
    C = d.n_levels * d.n_features
    out = np.empty((x.shape[0], C), dtype=np.float32)
    dydx = np.empty((x.shape[0], C, 3), dtype=np.float32) if want_dydx else None
    clib().orc_hashgrid_fwd(ctypes.byref(d), _fptr(params), _fptr(x), x.shape[0], _fptr(out),
                            _fptr(dydx) if want_dydx else None)
    return out, dydx


def hashgrid_bwd_params(d, x, dy):
    x = np.ascontiguousarray(x, dtype=np.float32)
    dy = np.ascontiguousarray(dy, dtype=np.float32)
    g = np.empty(d.n_params, dtype=np.float32)
    clib().orc_hashgrid_bwd_params(ctypes.byref(d), _fptr(x), _fptr(dy), x.shape[0], _fptr(g))
    return g


def hashgrid_bwd_input(dy, dydx):
    dy = np.ascontiguousarray(dy, dtype=np.float32)
    dydx = np.ascontiguousarray(dydx, dtype=np.float32)
    out = np.empty((dy.shape[0], 3), dtype=np.float32)
    clib().orc_hashgrid_bwd_input(_fptr(dy), _fptr(dydx), dy.shape[0], dy.shape[1], _fptr(out))
    return out


# --------------------------------------------------------------------------------------
# hash grid: independent numpy restatement (small N; cross-checks the C file)
# --------------------------------------------------------------------------------------
_PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


def _libm_f32(name):
    f = getattr(ctypes.CDLL("libm.so.6"), name)
    f.argtypes = [ctypes.c_float]; f.restype = ctypes.c_float
    return f


def np_level_tables(n_levels, log2_hashmap_size, base_resolution, pls):
    """
    tcnn grid.h grid_scale / grid_resolution / offset table in float32 like the original.  exp2f/log2f come
    from libm through ctypes (numpy's own float32 exp2/log2 loops round differently in the last place).
    """
    exp2f, log2f = _libm_f32("exp2f"), _libm_f32("log2f")
    log2_pls = np.float32(log2f(np.float32(pls)))
    scales, ress, offs = [], [], [0]
    for l in range(n_levels):
        s = np.float32(np.float32(exp2f(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0))
        r = int(np.ceil(s)) + 1
        n = r ** 3
        n = min(n, (2 ** 32 - 1) // 2)
        n = ((n + 7) // 8) * 8
        n = min(n, 1 << log2_hashmap_size)
        scales.append(s); ress.append(r); offs.append(offs[-1] + n)
    return np.array(scales, np.float32), np.array(ress, np.uint32), np.array(offs, np.uint32)


def np_grid_index(hs, res, g):
    """tcnn grid.h grid_index<3,CoherentPrime>; g: uint32 [...,3]."""
    with np.errstate(over="ignore"):
        stride = np.uint64(1); dense = np.zeros(g.shape[:-1], np.uint32); ndim = 0
        for dim in range(3):
            if stride <= hs:
                dense = dense + g[..., dim] * np.uint32(stride & np.uint64(0xFFFFFFFF))
                stride = np.uint64((int(stride) * int(res)) & 0xFFFFFFFF)  # uint32 arithmetic like the original
                ndim += 1
        if hs < stride:
            idx = (g[..., 0] * _PRIMES[0]) ^ (g[..., 1] * _PRIMES[1]) ^ (g[..., 2] * _PRIMES[2])
        else:
            idx = dense
    return (idx % np.uint32(hs)).astype(np.uint32)


def np_hashgrid_fwd(params, x, n_levels, n_features, log2_hashmap_size, base_resolution, pls):
    """tcnn kernel_grid in numpy float32 (fma emulated in float64 then rounded: exact for one product+sum)."""
    scales, ress, offs = np_level_tables(n_levels, log2_hashmap_size, base_resolution, pls)
    x = np.asarray(x, np.float32); N = x.shape[0]; Fe = n_features
    out = np.zeros((N, n_levels * Fe), np.float32)
    idx_all = np.zeros((N, n_levels, 8), np.uint32)
    for l in range(n_levels):
        hs = int(offs[l + 1] - offs[l]); grid = params[int(offs[l]) * Fe:int(offs[l + 1]) * Fe].reshape(hs, Fe)
        p = (x.astype(np.float64) * np.float64(scales[l]) + 0.5).astype(np.float32)  # fmaf
        fl = np.floor(p); g = fl.astype(np.int64).astype(np.uint32); pos = (p - fl).astype(np.float32)
        res = np.zeros((N, Fe), np.float32)
        for c in range(8):
            w = np.ones(N, np.float32); gl = g.copy()
            for k in range(3):
                if (c >> k) & 1:
                    w = w * pos[:, k]; gl[:, k] = g[:, k] + np.uint32(1)
                else:
                    w = w * (np.float32(1.0) - pos[:, k])
            idx = np_grid_index(hs, int(ress[l]), gl); idx_all[:, l, c] = idx
            res = (w[:, None].astype(np.float64) * grid[idx].astype(np.float64) + res.astype(np.float64)).astype(np.float32)
        out[:, l * Fe:(l + 1) * Fe] = res
    return out, idx_all


# --------------------------------------------------------------------------------------
# torch-CPU module with the tcnn.Encoding call contract (src/UNISLAM.py:242-254, decoders.py:103)
# --------------------------------------------------------------------------------------
class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, desc):
        xn = x.detach().contiguous().numpy().astype(np.float32)
        need_dx = x.requires_grad
        out, dydx = hashgrid_fwd(desc, params.detach().contiguous().numpy(), xn, want_dydx=need_dx)
        ctx.desc = desc; ctx.xn = xn; ctx.dydx = dydx
        ctx.need = (x.requires_grad, params.requires_grad)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, dy):
        dyn = dy.contiguous().numpy().astype(np.float32)
        gx = gp = None
        if ctx.need[0]:
            gx = torch.from_numpy(hashgrid_bwd_input(dyn, ctx.dydx))
        if ctx.need[1]:
            gp = torch.from_numpy(hashgrid_bwd_params(ctx.desc, ctx.xn, dyn))
        return gx, gp, None


class HashGridOracle(nn.Module):
    """CPU stand-in honouring the tcnn.Encoding contract: enc(x[N,3] in [0,1]) -> [N, L*F] fp32; .params flat fp32."""

    def __init__(self, n_input_dims=3, encoding_config=None, dtype=torch.float, seed=1337):
        super().__init__()
        c = encoding_config
        assert n_input_dims == 3 and c["otype"] == "HashGrid"
        self.desc = make_grid_desc(c["n_levels"], c["n_features_per_level"], c["log2_hashmap_size"],
                                   c["base_resolution"], c["per_level_scale"])
        self.n_output_dims = c["n_levels"] * c["n_features_per_level"]
        g = torch.Generator().manual_seed(seed)
        # tcnn initialises grid params U(-1e-4, 1e-4) (pcg32 stream; not reproduced bit for bit)
        self.params = nn.Parameter((torch.rand(self.desc.n_params, generator=g) * 2 - 1) * 1e-4)

    def forward(self, x):
        return _HashGridFn.apply(x, self.params, self.desc)


# --------------------------------------------------------------------------------------
# decoders (src/networks/decoders.py:24-205)
# --------------------------------------------------------------------------------------
def mlp_forward(h, weights, biases, out_act):
    """decoders.py:125-128 / :150-153 (torch path) and the tcnn FullyFusedMLP math in fp32 (biases=None)."""
    n = len(weights)
    for i, W in enumerate(weights):
        h = h @ W.t()
        if biases is not None and biases[i] is not None:
            h = h + biases[i]
        if i < n - 1:
            h = torch.relu(h)
    if out_act == "tanh":
        return torch.tanh(h)
    if out_act == "sigmoid":
        return torch.sigmoid(h)
    return h


def tcnn_mlp_unpack(params, n_in, width, n_hidden, n_out):
    """
    tiny-cuda-nn FullyFusedMLP parameter layout [tcnn-upstream]: row-major [out,in] matrices back to back:
    first (width x n_in), (n_hidden-1) hidden (width x width), last (16-padded n_out x width).
    """
    pad_out = ((n_out + 15) // 16) * 16
    ws, o = [], 0
    ws.append(params[o:o + width * n_in].view(width, n_in)); o += width * n_in
    for _ in range(n_hidden - 1):
        ws.append(params[o:o + width * width].view(width, width)); o += width * width
    ws.append(params[o:o + pad_out * width].view(pad_out, width)[:n_out]); o += pad_out * width
    assert o == params.numel()
    return ws


class DecodersOracle(nn.Module):
    """
    Restatement of reference Decoders (decoders.py:35-205) for grid_mode == 'hash_grid'.
    tcnn_network=False: nn.Linear stacks with bias (state_dict keys identical to the reference).
    tcnn_network=True : two flat fp32 'params' vectors in the FullyFusedMLP layout, evaluated in fp32.
    """

    def __init__(self, c_dim=32, hidden_size=16, n_blocks=2, learnable_beta=True, tcnn_network=False):
        super().__init__()
        self.c_dim, self.hidden_size, self.n_blocks, self.tcnn_network = c_dim, hidden_size, n_blocks, tcnn_network
        if tcnn_network:
            n_hidden = n_blocks - 1
            n = hidden_size * c_dim + (n_hidden - 1) * hidden_size ** 2 + 16 * hidden_size
            self.sdf_params = nn.Parameter(torch.empty(n).uniform_(-0.3, 0.3))
            self.color_params = nn.Parameter(torch.empty(n).uniform_(-0.3, 0.3))
        else:
            self.linears = nn.ModuleList([nn.Linear(c_dim, hidden_size)] +
                                         [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
            self.c_linears = nn.ModuleList([nn.Linear(c_dim, hidden_size)] +
                                           [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
            self.output_linear = nn.Linear(hidden_size, 1)
            self.c_output_linear = nn.Linear(hidden_size, 3)
        self.beta = nn.Parameter(10 * torch.ones(1)) if learnable_beta else 10

    def _feat(self, p_nor, grids):
        return grids[0](torch.clamp(p_nor, min=0, max=1))          # decoders.py:101-103

    def get_raw_sdf(self, p_nor, scene_rep):                        # decoders.py:107-130
        h = self._feat(p_nor, scene_rep[0])
        if self.tcnn_network:
            ws = tcnn_mlp_unpack(self.sdf_params, self.c_dim, self.hidden_size, self.n_blocks - 1, 1)
            return mlp_forward(h, ws, None, "tanh").squeeze()
        ws = [l.weight for l in self.linears] + [self.output_linear.weight]
        bs = [l.bias for l in self.linears] + [self.output_linear.bias]
        return mlp_forward(h, ws, bs, "tanh").squeeze()

    def get_raw_rgb(self, p_nor, scene_rep):                        # decoders.py:132-155
        h = self._feat(p_nor, scene_rep[1])
        if self.tcnn_network:
            ws = tcnn_mlp_unpack(self.color_params, self.c_dim, self.hidden_size, self.n_blocks - 1, 3)
            return mlp_forward(h, ws, None, "sigmoid")
        ws = [l.weight for l in self.c_linears] + [self.c_output_linear.weight]
        bs = [l.bias for l in self.c_linears] + [self.c_output_linear.bias]
        return mlp_forward(h, ws, bs, "sigmoid")

    def forward(self, p, scene_rep):                                # decoders.py:182-205
        shp = p.shape
        p_nor = p.reshape(-1, 3)
        sdf = self.get_raw_sdf(p_nor, scene_rep)
        rgb = self.get_raw_rgb(p_nor, scene_rep)
        return torch.cat([rgb, sdf.unsqueeze(-1)], dim=-1).reshape(*shp[:-1], -1)


# --------------------------------------------------------------------------------------
# rays (src/common.py:35-46, 95-180, 210-228)
# --------------------------------------------------------------------------------------
def camera_dirs(i, j, fx, fy, cx, cy):
    """OpenGL pinhole directions, common.py:41 / :100 / :221-222."""
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def get_camera_rays(H, W, fx, fy, cx, cy):
    """common.py:35-46 (cached per dataset at datasets.py:134-135)."""
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    return camera_dirs(i, j, fx, fy, cx, cy)


def rotate_dirs(dirs, R):
    """common.py:104 / :161 / :226: sum(dirs[..., None, :] * R, -1) == R @ dir."""
    return torch.sum(dirs.unsqueeze(-2) * R, -1)


def get_rays(H, W, fx, fy, cx, cy, c2w):
    """common.py:210-228: whole-image rays."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i, j = i.t(), j.t()
    rays_d = rotate_dirs(camera_dirs(i, j, fx, fy, cx, cy), c2w[:3, :3])
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_samples_all(n, c2ws, depths, colors, rays_d_pool, indices=None):
    """
    common.py:152-166 (mapping): per-frame pools depths[b,P], colors[b,P,3], rays_d_pool[b,P,3].
    `indices` [b,n] int64 replaces the torch.randint draw at :155 when given.
    """
    b, P = depths.shape
    if indices is None:
        indices = torch.randint(P, (n * b,)).reshape(b, -1)
    sd = torch.gather(depths, 1, indices)
    sc = torch.gather(colors, 1, indices.unsqueeze(-1).expand(-1, -1, 3))
    rd_all = rotate_dirs(rays_d_pool, c2ws[:, None, :3, :3])
    ro_all = c2ws[:, None, :3, -1].expand(rd_all.shape)
    gi = indices.unsqueeze(-1).expand(-1, -1, 3)
    rd = torch.gather(rd_all, 1, gi); ro = torch.gather(ro_all, 1, gi)
    return ro.reshape(-1, 3), rd.reshape(-1, 3), sd.reshape(-1), sc.reshape(-1, 3)


def get_samples(H0, H1, W0, W1, n, fx, fy, cx, cy, c2ws, depths, colors, indices=None):
    """
    common.py:168-180 -> :133-150 -> :109-131 -> :95-107 (tracking): crop, pick n*b pixels, build rays.
    `indices` [n*b] int64 replaces torch.randint at :116.
    """
    b = c2ws.shape[0]
    if not (H0 == 0 and W0 == 0):
        depths = depths[:, H0:H1, W0:W1]; colors = colors[:, H0:H1, W0:W1]
    i, j = torch.meshgrid(torch.linspace(W0, W1 - 1, W1 - W0), torch.linspace(H0, H1 - 1, H1 - H0), indexing="ij")
    i, j = i.t().reshape(-1), j.t().reshape(-1)
    if indices is None:
        indices = torch.randint(i.shape[0], (n * b,))
    indices = indices.clamp(0, i.shape[0])
    ii, jj = i[indices].reshape(b, -1), j[indices].reshape(b, -1)
    idx = indices.reshape(b, -1)
    sd = torch.gather(depths.reshape(b, -1), 1, idx)
    sc = torch.gather(colors.reshape(b, -1, 3), 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    rd = rotate_dirs(camera_dirs(ii, jj, fx, fy, cx, cy), c2ws[:, None, :3, :3])
    ro = c2ws[:, None, :3, -1].expand(rd.shape)
    return ro.reshape(-1, 3), rd.reshape(-1, 3), sd.reshape(-1), sc.reshape(-1, 3)


def bbox_far(rays_o, rays_d, bound):
    """Mapper.py:396-402 / Tracker.py:177-183 / Renderer.py:108-111: far = min_dim max_side (bound - o)/d."""
    t = (bound.unsqueeze(0) - rays_o.unsqueeze(-1)) / rays_d.unsqueeze(-1)
    far, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    return far


# --------------------------------------------------------------------------------------
# quaternion pose helpers (common.py:182-208; pytorch3d real-first convention, restated)
# --------------------------------------------------------------------------------------
def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def cam_pose_to_matrix(poses):
    """common.py:196-208: [quat(4, real first), t(3)] -> 4x4."""
    c2w = torch.eye(4).unsqueeze(0).repeat(poses.shape[0], 1, 1)
    c2w[:, :3, :3] = quaternion_to_matrix(poses[:, :4])
    c2w[:, :3, 3] = poses[:, 4:]
    return c2w


def matrix_to_quaternion(matrix):
    """pytorch3d.transforms.matrix_to_quaternion (pinned commit 47d5dc88, requirements.txt:91; absent here: restated, parity unpinned):
    real part first; the candidate built from the largest of the four |q_i| is taken, denominators floored at 0.1."""
    m = matrix.reshape(matrix.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, dim=-1)
    x = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0.0)), torch.zeros_like(x))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)


def matrix_to_cam_pose(mats):
    """common.py:182-194 (RT=True): 4x4 -> [quat(4, real first), t(3)]."""
    return torch.cat([matrix_to_quaternion(mats[:, :3, :3]), mats[:, :3, 3]], dim=-1)


# --------------------------------------------------------------------------------------
# renderer (src/utils/Renderer.py:42-158) and sample_pdf (src/common.py:49-85)
# --------------------------------------------------------------------------------------
def perturbation(z_vals, t_rand=None):
    """Renderer.py:42-57."""
    mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    if t_rand is None:
        t_rand = torch.rand(z_vals.shape)
    return lower + (upper - lower) * t_rand


def sample_z_with_depth(gt_nonzero, truncation, n_stratified, n_importance, perturb, t_rand=None):
    """Renderer.py:83-100 for rays with gt depth > 0.  gt_nonzero: [R,1]."""
    t_uni = torch.linspace(0., 1., steps=n_stratified)
    t_surf = torch.linspace(0., 1., steps=n_importance)
    z_surf = gt_nonzero.expand(-1, n_importance) - (1.5 * truncation) + (3 * truncation * t_surf)
    z_free = 0.0 + 1.2 * gt_nonzero.expand(-1, n_stratified) * t_uni
    z, _ = torch.sort(torch.cat([z_free, z_surf], dim=-1), dim=-1)
    if perturb:
        z = perturbation(z, t_rand)
    return z


def sample_pdf(bins, weights, n_samples, u=None):
    """common.py:49-85 including the quirk at :55-56 (pdf is NOT normalised)."""
    cdf = torch.cumsum(weights, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b = torch.gather(cdf, 1, below); cdf_a = torch.gather(cdf, 1, above)
    bin_b = torch.gather(bins, 1, below); bin_a = torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def sdf2alpha(sdf, beta=10):
    """Renderer.py:154-158."""
    return 1. - torch.exp(-beta * torch.sigmoid(-sdf * beta))


def alpha_to_weights(alpha):
    """Renderer.py:141-142 (note the +1e-10 inside the cumprod)."""
    ones = torch.ones((alpha.shape[0], 1))
    return alpha * torch.cumprod(torch.cat([ones, (1. - alpha + 1e-10)], -1), -1)[:, :-1]


def composite(raw, z_vals, beta):
    """Renderer.py:140-152: returns the reference's 7-tuple."""
    alpha = sdf2alpha(raw[..., 3], beta)
    w = alpha_to_weights(alpha)
    rgb = torch.sum(w[..., None] * raw[..., :3], -2)
    depth = torch.sum(w * z_vals, -1)
    term = torch.sum(w, -1)
    pixel_unc = torch.square(1 - torch.sum(w, -1))
    depth_unc = torch.sqrt(torch.sum(w * (depth[..., None] - z_vals) ** 2, -1))
    return term, pixel_unc, depth, rgb, raw[..., 3], z_vals, depth_unc


def normalize_3d_coordinate(p, bound):
    """common.py:231-245: to [-1,1] (only used by the zero-depth coarse pass, Renderer.py:120)."""
    p = p.reshape(-1, 3)
    return torch.stack([((p[:, k] - bound[k, 0]) / (bound[k, 1] - bound[k, 0])) * 2 - 1.0 for k in range(3)], -1)


def render_batch_ray(scene_rep, decoders, rays_d, rays_o, truncation, gt_depth, bound, n_stratified, n_importance,
                     perturb=True, rand=None):
    """
    Renderer.py:59-152.  `rand` may carry pre-drawn uniforms {'z': [R1,S], 'z_uni': [R0,n_strat], 'u': [R0,n_imp]};
    missing entries are drawn with torch.rand in the reference's own call order, so seeding the
    global CPU generator reproduces the reference's stream.
    """
    rand = rand or {}
    n_rays = rays_o.shape[0]
    S = n_stratified + n_importance
    z_vals = torch.empty([n_rays, S])
    gt_depth = gt_depth.reshape(-1, 1)
    gt_mask = (gt_depth > 0).squeeze(-1)
    z_vals[gt_mask] = sample_z_with_depth(gt_depth[gt_mask], truncation, n_stratified, n_importance, perturb,
                                          rand.get("z"))
    if not gt_mask.all():
        with torch.no_grad():                                       # Renderer.py:104-130
            ro, rd = rays_o[~gt_mask].detach(), rays_d[~gt_mask].detach()
            far = bbox_far(ro, rd, bound).unsqueeze(-1) + 0.01
            t_uni = torch.linspace(0., 1., steps=n_stratified)
            z_uni = 0.0 * (1. - t_uni) + far * t_uni
            if perturb:
                z_uni = perturbation(z_uni, rand.get("z_uni"))
            pts_uni = ro.unsqueeze(1) + rd.unsqueeze(1) * z_uni.unsqueeze(-1)
            sdf_uni = decoders.get_raw_sdf(normalize_3d_coordinate(pts_uni.clone(), bound), scene_rep)
            sdf_uni = sdf_uni.reshape(*pts_uni.shape[0:2])
            w_uni = alpha_to_weights(sdf2alpha(sdf_uni, decoders.beta))
            mid = .5 * (z_uni[..., 1:] + z_uni[..., :-1])
            z_s = sample_pdf(mid, w_uni[..., 1:-1], n_importance, rand.get("u"))
            z_uni, _ = torch.sort(torch.cat([z_uni, z_s], -1), -1)
            z_vals[~gt_mask] = z_uni
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    pts = (pts - bound[:, 0]) / (bound[:, 1] - bound[:, 0])        # Renderer.py:137
    raw = decoders(pts, scene_rep)
    return composite(raw, z_vals, decoders.beta)


# --------------------------------------------------------------------------------------
# losses (src/Mapper.py:141-175 == src/Tracker.py:113-147; Mapper.py:414-440; Tracker.py:208-238)
# --------------------------------------------------------------------------------------
def sdf_loss_masks(z_vals, gt_depth, truncation):
    g = gt_depth[:, None]
    front = z_vals < (g - truncation)
    back = z_vals > (g + truncation)
    center = (z_vals > (g - 0.4 * truncation)) & (z_vals < (g + 0.4 * truncation))
    tail = (~front) & (~back) & (~center)
    return front, back, center, tail


def sdf_losses(sdf, z_vals, gt_depth, truncation, w_fs, w_center, w_tail):
    """Mapper.py:141-175.  Empty masks yield NaN exactly like torch.mean of an empty tensor."""
    front, _, center, tail = sdf_loss_masks(z_vals, gt_depth, truncation)
    g = gt_depth[:, None].expand(z_vals.shape)
    fs = torch.mean(torch.square(sdf[front] - 1.0))
    pred = z_vals + sdf * truncation
    ce = torch.mean(torch.square(pred[center] - g[center]))
    ta = torch.mean(torch.square(pred[tail] - g[tail]))
    return w_fs * fs + w_center * ce + w_tail * ta


def mapping_loss(ret, gt_depth, gt_color, truncation, w, mask_mode="original"):
    """Mapper.py:411-440.  w = dict(fs, center, tail, color, depth)."""
    _, pixel_unc, depth, color, sdf, z_vals, _ = ret
    alpha_mask = (1 - pixel_unc.detach()) > 0.99
    depth_mask = (gt_depth > 0) & alpha_mask
    if mask_mode == "original":
        loss = sdf_losses(sdf[depth_mask], z_vals[depth_mask], gt_depth[depth_mask], truncation, w["fs"], w["center"], w["tail"])
        loss = loss + w["color"] * torch.square(gt_color - color).mean()
        loss = loss + w["depth"] * torch.square(gt_depth[depth_mask] - depth[depth_mask]).mean()
    else:
        loss = sdf_losses(sdf, z_vals, gt_depth, truncation, w["fs"], w["center"], w["tail"])
        loss = loss + w["color"] * torch.square(gt_color - color).mean()
        loss = loss + w["depth"] * torch.square(gt_depth - depth).mean()
    return loss


def tracking_loss(ret, gt_depth, gt_color, truncation, w, mask_mode="original"):
    """Tracker.py:206-238."""
    _, pixel_unc, depth, color, sdf, z_vals, _ = ret
    alpha_mask = (1 - pixel_unc.detach()) > 0.99
    err = (gt_depth - depth.detach()).abs()
    depth_mask = (err < 10 * err.median()) & alpha_mask
    if mask_mode == "original":
        loss = sdf_losses(sdf[depth_mask], z_vals[depth_mask], gt_depth[depth_mask], truncation, w["fs"], w["center"], w["tail"])
        loss = loss + w["color"] * torch.square(gt_color - color)[depth_mask].mean()
        loss = loss + w["depth"] * torch.square(gt_depth[depth_mask] - depth[depth_mask]).mean()
    else:
        loss = sdf_losses(sdf, z_vals, gt_depth, truncation, w["fs"], w["center"], w["tail"])
        loss = loss + (w["color"] * torch.square(gt_color - color)).mean()
        loss = loss + (w["depth"] * torch.square(gt_depth - depth)).mean()
    return loss


# --------------------------------------------------------------------------------------
# one full mapping iteration on CPU (Mapper.py:366-445) -- used as bench.py's cpu_baseline ("port")
# --------------------------------------------------------------------------------------
def window_rays(c2w_first, cam_poses, depths, colors, dirs_pool, indices, extra=None, indices_extra=None):
    """Mapper.py:372-393 for a window whose poses 1.. are the 7-vectors `cam_poses` (a leaf the caller differentiates): the rays of
    get_samples_all over all frames, then -- extra = (n_frames, n) -- of a second call over the newest n_frames frames."""
    c2ws = torch.cat([c2w_first[None], cam_pose_to_matrix(cam_poses)], dim=0) if cam_poses is not None else c2w_first[None]
    out = list(get_samples_all(indices.shape[1], c2ws, depths, colors, dirs_pool, indices))
    if extra is not None:
        nf = extra[0]
        o2 = get_samples_all(extra[1], c2ws[-nf:], depths[-nf:], colors[-nf:], dirs_pool[-nf:], indices_extra)
        out = [torch.cat([a, b], dim=0) for a, b in zip(out, o2)]
    return out


def mapping_iteration(scene_rep, decoders, optimizer, rays_o, rays_d, gt_depth, gt_color, bound, truncation,
                      n_stratified, n_importance, w, mask_mode="original", perturb=True):
    with torch.no_grad():
        inside = bbox_far(rays_o, rays_d, bound) >= gt_depth
    rays_o, rays_d, gt_depth, gt_color = rays_o[inside], rays_d[inside], gt_depth[inside], gt_color[inside]
    ret = render_batch_ray(scene_rep, decoders, rays_d, rays_o, truncation, gt_depth, bound, n_stratified,
                           n_importance, perturb)
    loss = mapping_loss(ret, gt_depth, gt_color, truncation, w, mask_mode)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()
