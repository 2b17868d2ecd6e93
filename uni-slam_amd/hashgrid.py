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
    """
    features = enc(x) as ONE module call -- the seam `import unislam_amd.tcnn as tcnn` alone gives the reference (src/UNISLAM.py:242-253,
    src/networks/decoders.py:103).  Where the table wants a gradient and the batch is one the binned kernels take (r6): the forward pass is
    us_hashgrid_fwd_counted (the binning counts ride along with the gathers) + us_hashgrid_bwd_scan (the gradient table is allocated here;
    the scan clears what it must), the backward pass us_hashgrid_bwd_binned(COUNTED | SCANNED) -- no recount, and the 0.34 GB scratch is
    cached on the module (`owner`), not allocated per call.  A forward pass overtaken by another one of the same module before its backward
    pass (the cached counts are then someone else's) counts again in a scratch of its own.
    """

    @staticmethod
    def forward(ctx, x, params, desc, flags, bwd_mode, owner=None, track=False):
        x = L.f32(x.detach())
        p = L.f32(params.detach())
        n = x.shape[0]
        C = desc.n_levels * desc.n_features
        out = torch.empty((n, C), dtype=torch.float32, device=x.device)
        lib, dp = L.lib(), ctypes.byref(desc)
        ctx.counted = None
        # the binned path addresses its 12-byte records with 32-bit offsets: beyond ~2.8 M points (L = 16, F = 2) fall back
        fits = n * 8 * desc.n_levels * (1 + desc.n_features) * 4 <= 0xFFFFFFFF
        binned = fits and (bwd_mode == 3 or (bwd_mode == -1 and n >= 16384))
        if track and owner is not None and binned and ctx.needs_input_grad[1] and lib.us_hashgrid_bwd_binned_supported(dp, n):
            ws, nbytes = owner._workspace(n)
            L.check(lib.us_hashgrid_fwd_counted(dp, L.ptr(p), L.ptr(x), n, L.ptr(out), flags, L.ptr(ws), nbytes, L.stream()), "us_hashgrid_fwd_counted")
            gp = torch.empty(desc.n_params, dtype=torch.float32, device=x.device)
            L.check(lib.us_hashgrid_bwd_scan(dp, n, L.ptr(gp), flags | L.US_GRID_BWD_OVERWRITE | owner.grid_bwd_flags, L.ptr(ws), nbytes, L.stream()),
                    "us_hashgrid_bwd_scan")
            owner._ws_gen += 1
            ctx.counted = (gp, ws, nbytes, owner._ws_gen, owner.grid_bwd_flags)
        else:
            L.check(lib.us_hashgrid_fwd(dp, L.ptr(p), L.ptr(x), n, L.ptr(out), None, flags, L.stream()), "us_hashgrid_fwd")
        ctx.desc, ctx.flags, ctx.bwd_mode, ctx.owner = desc, flags, bwd_mode, owner
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
        if ctx.needs_input_grad[1] and ctx.counted is not None:
            gp, ws, nbytes, gen, det = ctx.counted
            ctx.counted = None                # (the returned gradient must be the only reference: AccumulateGrad then takes it without a copy)
            flags = ctx.flags | L.US_GRID_BWD_OVERWRITE | det
            if gen == ctx.owner._ws_gen:
                flags |= L.US_GRID_BWD_COUNTED | L.US_GRID_BWD_SCANNED
            else:                             # another forward pass of this module has used the cached scratch since: count again, in a scratch of this call's own
                ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            L.check(L.lib().us_hashgrid_bwd_binned(ctypes.byref(desc), L.ptr(x), L.ptr(dy), n, L.ptr(gp), flags, L.ptr(ws), nbytes, L.stream()),
                    "us_hashgrid_bwd_binned")
        elif ctx.needs_input_grad[1]:
            mode = ctx.bwd_mode
            fits = n * 8 * desc.n_levels * (1 + desc.n_features) * 4 <= 0xFFFFFFFF
            binned = fits and (mode == 3 or (mode == -1 and n >= 16384))
            if mode == 3 and not fits:
                mode = 1
            gp = (torch.empty if binned else torch.zeros)(desc.n_params, dtype=torch.float32, device=x.device)
            if binned:
                # bin once, accumulate in f64 (csrc/hashgrid_binned.hip); scratch: the module's cached one where there is a module
                nbytes = int(L.lib().us_hashgrid_bwd_workspace_bytes(ctypes.byref(desc), n))
                if ctx.owner is not None:
                    ws, nbytes = ctx.owner._workspace(n)
                    ctx.owner._ws_gen += 1    # (whatever counts a pending backward pass expects there are gone)
                else:
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
        return gx, gp, None, None, None, None, None


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
        self.grid_bwd_flags = 0         # e.g. US_GRID_BWD_DETERMINISTIC: no float atomics in the table gradient
        self._ws, self._ws_gen = None, 0   # cached scratch of the binned table gradient (never pickled) + its generation counter
        g = torch.Generator().manual_seed(seed)
        # tcnn initialises grid parameters U(-1e-4, 1e-4) from a pcg32 stream seeded 1337; same distribution here,
        # not the same bit stream (initialisation is not part of the hot path's parity contract)
        self.params = nn.Parameter((torch.rand(self.desc.n_params, generator=g) * 2 - 1) * 1e-4)

    # the ctypes descriptor is rebuilt on unpickle / deepcopy (Tracker.py:107-108, spawn at UNISLAM.py:295-298)
    def __getstate__(self):
        s = self.__dict__.copy()
        s.pop("desc", None)
        s["_ws"] = None
        return s

    def _workspace(self, n):
        """(scratch tensor, bytes) of the binned table gradient for n points, cached on the module and grown on demand"""
        nbytes = int(L.lib().us_hashgrid_bwd_workspace_bytes(ctypes.byref(self.desc), n))
        dev = self.params.device
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = None               # (free the old one first)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws_gen += 1             # whatever a pending backward pass expects to find is gone
        return self._ws, nbytes

    def __setstate__(self, s):
        self.__dict__.update(s)
        self.__dict__.setdefault("grid_bwd_flags", 0); self.__dict__.setdefault("_ws", None); self.__dict__.setdefault("_ws_gen", 0)
        c = self.encoding_config
        self.desc = make_grid_desc(int(c.get("n_levels", 16)), int(c.get("n_features_per_level", 2)),
                                   int(c.get("log2_hashmap_size", 19)), int(c.get("base_resolution", 16)),
                                   float(c.get("per_level_scale", 2.0)))

    def __deepcopy__(self, memo):
        new = HashGridEncoding(3, self.encoding_config)
        new.bwd_mode, new.clamp_input, new.grid_bwd_flags = self.bwd_mode, self.clamp_input, self.grid_bwd_flags
        new.params = nn.Parameter(self.params.detach().clone(), requires_grad=self.params.requires_grad)   # same device
        return new

    def forward(self, x, clamp=None):
        if x.dim() != 2 or x.shape[1] != 3:
            raise ValueError(f"HashGridEncoding: expected [N,3] positions, got {tuple(x.shape)}")
        clamp = self.clamp_input if clamp is None else clamp
        return _HashGridFn.apply(x, self.params, self.desc, L.US_GRID_CLAMP01 if clamp else 0, self.bwd_mode, self,
                                 torch.is_grad_enabled() and self.params.requires_grad)
