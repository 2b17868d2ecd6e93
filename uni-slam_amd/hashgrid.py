"""
HashGridEncoding -- drop-in for tcnn.Encoding(otype="HashGrid") as Uni-SLAM constructs it
(reference src/UNISLAM.py:242-253) and calls it (src/networks/decoders.py:103):

    enc = HashGridEncoding(n_input_dims=3, encoding_config={...}, dtype=torch.float)
    feat = enc(x)            # x [N,3] in [0,1]  ->  [N, n_levels*n_features] fp32
    enc.params               # ONE flat fp32 nn.Parameter (given directly to Adam, src/Mapper.py:118-126)
    enc.n_output_dims

Forward/backward run the HIP kernels of csrc/hashgrid.hip through the C ABI; there is no CPU path.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L


def make_grid_desc(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale):
    d = L.GridDesc()
    L.check(L.lib().us_grid_desc_init(ctypes.byref(d), n_levels, n_features, log2_hashmap_size, base_resolution,
                                      ctypes.c_float(per_level_scale)), "us_grid_desc_init")
    return d


def grid_indices(desc, x, clamp=False):
    """[N, L, 8] corner entry indices (int64 view of the kernel's uint32) -- parity/debug helper"""
    x = L.f32(x)
    n = x.shape[0]
    idx = torch.empty((n, desc.n_levels, 8), dtype=torch.int32, device=x.device)
    L.check(L.lib().us_hashgrid_indices(ctypes.byref(desc), L.ptr(x), n, L.ptr(idx), int(clamp), L.stream()),
            "us_hashgrid_indices")
    return idx.to(torch.int64) & 0xFFFFFFFF


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, desc, flags, bwd_mode):
        x = L.f32(x.detach())
        p = L.f32(params.detach())
        n = x.shape[0]
        C = desc.n_levels * desc.n_features
        out = torch.empty((n, C), dtype=torch.float32, device=x.device)
        L.check(L.lib().us_hashgrid_fwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), n, L.ptr(out), None, flags,
                                        L.stream()), "us_hashgrid_fwd")
        ctx.desc, ctx.flags, ctx.bwd_mode = desc, flags, bwd_mode
        ctx.save_for_backward(x, p)           # no dy/dx tensor: the input gradient gathers the vertices again
        return out

    @staticmethod
    def backward(ctx, dy):
        x, p = ctx.saved_tensors
        desc = ctx.desc
        dy = L.f32(dy)
        n = x.shape[0]
        gx = gp = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((n, 3), dtype=torch.float32, device=x.device)
            L.check(L.lib().us_hashgrid_bwd_input_gather(ctypes.byref(desc), L.ptr(p), L.ptr(x), L.ptr(dy), n, L.ptr(gx), ctx.flags,
                                                         L.stream()), "us_hashgrid_bwd_input_gather")
        if ctx.needs_input_grad[1]:
            mode = ctx.bwd_mode
            # the binned path addresses its 12-byte records with 32-bit offsets: beyond ~2.8 M points (L = 16, F = 2) fall back
            fits = n * 8 * desc.n_levels * (1 + desc.n_features) * 4 <= 0xFFFFFFFF
            binned = fits and (mode == 3 or (mode == -1 and n >= 16384))
            if mode == 3 and not fits:
                mode = 1
            gp = (torch.empty if binned else torch.zeros)(desc.n_params, dtype=torch.float32, device=x.device)
            if binned:
                # bin once, accumulate in f64 (csrc/hashgrid_binned.hip); scratch comes from torch's caching allocator
                nbytes = int(L.lib().us_hashgrid_bwd_workspace_bytes(ctypes.byref(desc), n))
                ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                rc = L.lib().us_hashgrid_bwd_binned(ctypes.byref(desc), L.ptr(x), L.ptr(dy), n, L.ptr(gp),
                                                    ctx.flags | L.US_GRID_BWD_OVERWRITE, L.ptr(ws), nbytes, L.stream())
                if rc == L.US_ERR_CONFIG and mode == -1:
                    binned = False          # a table too large for the bin budget: the sliced kernels take any size
                    gp.zero_()
                else:
                    L.check(rc, "us_hashgrid_bwd_binned")
            if not binned:
                # the sliced kernel streams one level at a time: hand it level-major planes [L][N][F]
                dy_lm = dy.view(n, desc.n_levels, desc.n_features).permute(1, 0, 2).contiguous()
                L.check(L.lib().us_hashgrid_bwd_params(ctypes.byref(desc), L.ptr(x), L.ptr(dy_lm), n, L.ptr(gp), mode,
                                                       ctx.flags | L.US_GRID_LEVEL_MAJOR, L.stream()), "us_hashgrid_bwd_params")
        return gx, gp, None, None, None


class HashGridEncoding(nn.Module):
    def __init__(self, n_input_dims=3, encoding_config=None, dtype=torch.float, seed=1337):
        super().__init__()
        c = dict(encoding_config or {})
        if n_input_dims != 3:
            raise ValueError("HashGridEncoding: only 3-D inputs are supported (Uni-SLAM encodes xyz)")
        if c.get("otype", "HashGrid") not in ("HashGrid", "Grid"):
            raise ValueError(f"HashGridEncoding: unsupported otype {c.get('otype')}")
        if dtype not in (torch.float, torch.float32):
            raise ValueError("HashGridEncoding: fp32 tables only (the reference passes dtype=torch.float)")
        self.encoding_config = c
        self.n_input_dims = 3
        self.desc = make_grid_desc(int(c.get("n_levels", 16)), int(c.get("n_features_per_level", 2)),
                                   int(c.get("log2_hashmap_size", 19)), int(c.get("base_resolution", 16)),
                                   float(c.get("per_level_scale", 2.0)))
        self.n_output_dims = self.desc.n_levels * self.desc.n_features
        self.bwd_mode = -1              # -1 auto | 0 global atomics | 1 LDS slices | 2 slices+compaction | 3 binned f64
        self.clamp_input = False        # Decoders sets this to fold its torch.clamp(p, 0, 1) into the kernel
        g = torch.Generator().manual_seed(seed)
        # tcnn initialises grid parameters U(-1e-4, 1e-4) from a pcg32 stream seeded 1337; same distribution here,
        # not the same bit stream (initialisation is not part of the hot path's parity contract)
        self.params = nn.Parameter((torch.rand(self.desc.n_params, generator=g) * 2 - 1) * 1e-4)

    # the ctypes descriptor is rebuilt on unpickle / deepcopy (Tracker.py:107-108, spawn at UNISLAM.py:295-298)
    def __getstate__(self):
        s = self.__dict__.copy()
        s.pop("desc", None)
        return s

    def __setstate__(self, s):
        self.__dict__.update(s)
        c = self.encoding_config
        self.desc = make_grid_desc(int(c.get("n_levels", 16)), int(c.get("n_features_per_level", 2)),
                                   int(c.get("log2_hashmap_size", 19)), int(c.get("base_resolution", 16)),
                                   float(c.get("per_level_scale", 2.0)))

    def __deepcopy__(self, memo):
        new = HashGridEncoding(3, self.encoding_config)
        new.bwd_mode, new.clamp_input = self.bwd_mode, self.clamp_input
        new.params = nn.Parameter(self.params.detach().clone(), requires_grad=self.params.requires_grad)   # same device
        return new

    def forward(self, x, clamp=None):
        if x.dim() != 2 or x.shape[1] != 3:
            raise ValueError(f"HashGridEncoding: expected [N,3] positions, got {tuple(x.shape)}")
        clamp = self.clamp_input if clamp is None else clamp
        return _HashGridFn.apply(x, self.params, self.desc, L.US_GRID_CLAMP01 if clamp else 0, self.bwd_mode)
