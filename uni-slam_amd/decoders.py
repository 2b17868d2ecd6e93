"""
Decoders -- SDF and colour decoders with the API and state_dict layout of reference src/networks/decoders.py
(class Decoders, :24-205), computing on the fused MFMA MLP kernel (csrc/mlp.hip) and the HIP hash grid.

  cfg['grid']['tcnn_network'] True : self.sdf_decoder / self.color_decoder are FusedMLP modules with tcnn's flat
        `params` (keys sdf_decoder.params, color_decoder.params), no bias, n_hidden_layers = n_blocks-1   (:49-70)
  cfg['grid']['tcnn_network'] False: nn.Linear stacks with bias (keys linears.*, c_linears.*, output_linear.*,
        c_output_linear.*) exactly as the reference (:72-84); their weights are packed into the kernel's flat
        layout on every call (a few hundred floats) so autograd returns per-layer gradients.

Decoders.forward (decoders.py:182-205) is ONE autograd node where both encoders are HashGridEncodings of one geometry
(Uni-SLAM's sdf / colour pair): the joint encoder (both tables per launch, level-major features, the table gradient's
binning counts riding along), the decoder pair, and in the backward pass the decoder pair's backward launch and the
joint binned table gradient -- the launches MapStep issues, behind the reference's call (_DecodersFusedFn below).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L
from .hashgrid import HashGridEncoding
from .network import FusedMLP, fused_mlp, make_mlp_desc, mlp_n_params


def _off(t, k):
    """device pointer of element k of a float tensor"""
    return ctypes.c_void_p(t.data_ptr() + 4 * k)


class _DecodersFusedFn(torch.autograd.Function):
    """
    raw[N,4] = (rgb, sdf) = Decoders.forward(p_nor) for two HashGridEncodings of one geometry (decoders.py:182-205 with
    :91-105,118-128,143-153 inside), as ONE autograd node on the kernels of the straight-line mapping step:

      forward   us_hashgrid_fwd_joint (both tables, level-major features; with table gradients wanted, the binning counts ride along)
                -> us_mlp_fwd_pair (or two us_mlp_fwd where the decoders do not pair) writing raw[N,4] in place
                -> us_hashgrid_joint_scan (depends on the counts only; the gradient tables are allocated here)
      backward  us_mlp_bwd_pair -> us_hashgrid_bwd_joint (COUNTED | SCANNED) [-> us_hashgrid_bwd_input_gather x 2 for dL/dp]

    Inputs after the three modules and the grad-mode flag: p_nor [N,3], the two tables, then the decoders' parameters in the order of
    Decoders._dec_params() (their VALUES are read from the packed vector Decoders._packed_params() keeps; the tensors are here so
    that autograd routes the gradients).  Scratch (the table gradient's workspace, the decoders' partial-row buffers) is cached on
    the Decoders module; a forward pass that is overtaken by another one before its backward pass runs (the cached counts are then
    someone else's) falls back to counting again in a scratch of its own.
    """

    @staticmethod
    def forward(ctx, dec, es, ec, track, p_nor, tab_s, tab_c, *dec_params):
        lib, st, P = L.lib(), L.stream(), L.ptr
        x = L.f32(p_nor.detach())
        n, dev = x.shape[0], x.device
        ts, tc = L.f32(tab_s.detach()), L.f32(tab_c.detach())
        ds, dc = ctypes.byref(es.desc), ctypes.byref(ec.desc)
        ms, mc = ctypes.byref(dec.mlp_descs()[0]), ctypes.byref(dec.mlp_descs()[1])
        ps, pc = dec._packed_params()
        need = ctx.needs_input_grad
        want_tab = bool(track and (need[5] or need[6]))       # (under no_grad needs_input_grad still reports the tensors' flags)
        lv = es.desc.n_levels
        feat_s = torch.empty(lv * n * 2, dtype=torch.float32, device=dev)
        feat_c = torch.empty_like(feat_s)
        raw = torch.empty((n, 4), dtype=torch.float32, device=dev)
        ws, nbytes = (dec._fused_workspace(es, ec, n) if want_tab else (None, 0))
        flags = L.US_GRID_CLAMP01 | L.US_GRID_LEVEL_MAJOR
        # (r6) split-bf16 decoders: the encoder writes the features as their hi / lo operand pairs (same bytes, same values), which only the
        # decoders' launches of this node read
        d_s, d_c = dec.mlp_descs()
        split = bool(getattr(dec, "feat_split", L.FEAT_SPLIT_DEFAULT) and d_s.precision == 1 and d_c.precision == 1 and d_s.n_in == 32 and d_c.n_in == 32)
        ctx.mflags = mfl = L.US_MLP_LEVEL_MAJOR | (L.US_MLP_IN_SPLIT_BF16 if split else 0)
        L.check(lib.us_hashgrid_fwd_joint(ds, dc, P(ts), P(tc), P(x), n, P(feat_s), P(feat_c), flags | (L.US_GRID_FEAT_SPLIT_BF16 if split else 0),
                                          P(ws), nbytes, st), "us_hashgrid_fwd_joint")
        ctx.pair = dec._pair_ok()
        if ctx.pair:
            L.check(lib.us_mlp_fwd_pair(ms, mc, ps, pc, P(feat_s), P(feat_c), n, _off(raw, 3), 4, P(raw), 4, mfl, st),
                    "us_mlp_fwd_pair")
        else:
            L.check(lib.us_mlp_fwd(ms, ps, P(feat_s), n, _off(raw, 3), 4, mfl, st), "us_mlp_fwd")
            L.check(lib.us_mlp_fwd(mc, pc, P(feat_c), n, P(raw), 4, mfl, st), "us_mlp_fwd")
        ctx.dec, ctx.es, ctx.ec, ctx.counted = dec, es, ec, None
        # the backward pass reads the decoders' weights from the packed vector as it is THEN (they are not copied per call): note their
        # versions, so that an in-place update between forward and backward raises as it would for a tensor autograd saved
        ctx.versions = [p._version for p in dec_params] if any(need[4:]) else None
        if want_tab:
            g_s = torch.empty(es.desc.n_params, dtype=torch.float32, device=dev)
            g_c = torch.empty(ec.desc.n_params, dtype=torch.float32, device=dev)
            L.check(lib.us_hashgrid_joint_scan(ds, dc, n, P(g_s), P(g_c), flags | L.US_GRID_BWD_OVERWRITE | dec.grid_bwd_flags, P(ws), nbytes, st),
                    "us_hashgrid_joint_scan")
            dec._fused_gen += 1
            ctx.counted = (g_s, g_c, ws, nbytes, dec._fused_gen)
        ctx.save_for_backward(x, ts, tc, feat_s, feat_c, raw)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        lib, st, P = L.lib(), L.stream(), L.ptr
        x, ts, tc, feat_s, feat_c, raw = ctx.saved_tensors
        dec, es, ec = ctx.dec, ctx.es, ctx.ec
        n, dev = x.shape[0], x.device
        if ctx.versions is not None:
            now = [p._version for p in dec._dec_params()]
            if now != ctx.versions:
                k = next(i for i, (a, b) in enumerate(zip(now, ctx.versions)) if a != b)
                raise L.UniSlamHipError(f"Decoders.forward's backward pass: decoder parameter {k} (shape {tuple(dec._dec_params()[k].shape)}) was modified in "
                                        f"place after the forward pass (version {ctx.versions[k]} -> {now[k]}, e.g. an optimizer.step() between forward and "
                                        "backward): its gradient would be taken at the new weights; run the forward pass again")
        d_raw = L.f32(d_raw)
        ds, dc = ctypes.byref(es.desc), ctypes.byref(ec.desc)
        ms, mc = ctypes.byref(dec.mlp_descs()[0]), ctypes.byref(dec.mlp_descs()[1])
        ps, pc = dec._packed_params()
        need = ctx.needs_input_grad
        want_x, want_tab, want_dec = bool(need[4]), bool(need[5] or need[6]), any(need[7:])
        d_feat_s, d_feat_c = torch.empty_like(feat_s), torch.empty_like(feat_c)
        n_s, n_c = dec._n_packed
        gflat = torch.zeros(n_s + n_c, dtype=torch.float32, device=dev) if want_dec else None
        mws_s, mws_c, mws_bytes = dec._mlp_workspaces(dev) if want_dec else (None, None, 0)
        gs_p, gc_p = (P(gflat), _off(gflat, n_s)) if want_dec else (None, None)
        if ctx.pair:
            L.check(lib.us_mlp_bwd_pair(ms, mc, ps, pc, P(feat_s), P(feat_c), _off(raw, 3), 4, P(raw), 4, _off(d_raw, 3), 4, P(d_raw), 4, n,
                                        P(d_feat_s), P(d_feat_c), gs_p, gc_p, ctx.mflags, P(mws_s), P(mws_c), mws_bytes, st),
                    "us_mlp_bwd_pair")
        else:
            L.check(lib.us_mlp_bwd(ms, ps, P(feat_s), _off(raw, 3), 4, _off(d_raw, 3), 4, n, P(d_feat_s), gs_p, ctx.mflags,
                                   P(mws_s), mws_bytes, st), "us_mlp_bwd")
            L.check(lib.us_mlp_bwd(mc, pc, P(feat_c), P(raw), 4, P(d_raw), 4, n, P(d_feat_c), gc_p, ctx.mflags,
                                   P(mws_c), mws_bytes, st), "us_mlp_bwd")
        g_s = g_c = None
        flags = L.US_GRID_CLAMP01 | L.US_GRID_LEVEL_MAJOR
        if want_tab:
            if ctx.counted is None:     # a second backward pass through this node (retain_graph): gradient tables and scratch of its own
                nbytes = dec._fused_bytes(es, ec, n)
                ctx.counted = (torch.empty(es.desc.n_params, dtype=torch.float32, device=dev), torch.empty(ec.desc.n_params, dtype=torch.float32, device=dev),
                               torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes, -1)
            g_s, g_c, ws, nbytes, gen = ctx.counted
            ctx.counted = None          # (the returned gradients must be the only references: AccumulateGrad then takes them without a copy)
            jflags = flags | L.US_GRID_BWD_OVERWRITE | dec.grid_bwd_flags
            if gen == dec._fused_gen:
                jflags |= L.US_GRID_BWD_COUNTED | L.US_GRID_BWD_SCANNED
            elif gen >= 0:              # another forward pass has used the cached scratch since: count again, in a scratch of this call's own
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            L.check(lib.us_hashgrid_bwd_joint(ds, dc, P(x), P(d_feat_s), P(d_feat_c), n, P(g_s), P(g_c), jflags, P(ws), nbytes, st),
                    "us_hashgrid_bwd_joint")
        gx = None
        if want_x:
            gx = torch.empty((n, 3), dtype=torch.float32, device=dev)
            L.check(lib.us_hashgrid_bwd_input_gather(ds, P(ts), P(x), P(d_feat_s), n, P(gx), flags, st), "us_hashgrid_bwd_input_gather")
            L.check(lib.us_hashgrid_bwd_input_gather(dc, P(tc), P(x), P(d_feat_c), n, P(gx), flags | L.US_GRID_ACCUMULATE, st),
                    "us_hashgrid_bwd_input_gather")
        dgrads = (None,) * (len(need) - 7)
        if want_dec:
            dgrads = tuple(gflat[o:o + p.numel()].view(p.shape) if nd else None
                           for (p, o), nd in zip(dec._pack_layout(), need[7:]))
        return (None, None, None, None, gx, g_s if need[5] else None, g_c if need[6] else None) + dgrads


class Decoders(nn.Module):
    def __init__(self, cfg, c_dim=32, hidden_size=16, truncation=0.08, n_blocks=2, learnable_beta=True):
        super().__init__()
        self.c_dim = c_dim
        self.cfg = cfg
        self.truncation = truncation
        self.n_blocks = n_blocks
        self.hidden_size = hidden_size
        self.tcnn_network = self.cfg['grid']['tcnn_network']
        # extension over the reference's yaml: MFMA operand type of the decoder MLPs ("fp32" | "bf16")
        self.mlp_precision = self.cfg.get('model', {}).get('mlp_precision', 'fp32') if hasattr(self.cfg, 'get') else 'fp32'
        if cfg['grid_mode'] != 'hash_grid':
            raise ValueError("Decoders: only grid_mode == 'hash_grid' exists in Uni-SLAM (decoders.py:116-120)")
        input_channels = c_dim
        if self.tcnn_network:
            mk = lambda n_out, act: FusedMLP(input_channels, n_out, {
                "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act,
                "n_neurons": hidden_size, "n_hidden_layers": n_blocks - 1, "precision": self.mlp_precision})
            self.sdf_decoder = mk(1, "Tanh")
            self.color_decoder = mk(3, "Sigmoid")
        else:
            self.linears = nn.ModuleList([nn.Linear(input_channels, hidden_size)] +
                                         [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
            self.c_linears = nn.ModuleList([nn.Linear(input_channels, hidden_size)] +
                                           [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
            self.output_linear = nn.Linear(hidden_size, 1)
            self.c_output_linear = nn.Linear(hidden_size, 3)
            if n_blocks not in (1, 2):
                raise ValueError("Decoders: the fused kernel covers n_blocks in {1, 2}")
            self._desc_sdf, self._desc_rgb = self.mlp_descs()
        if learnable_beta:
            self.beta = nn.Parameter(10 * torch.ones(1))
        else:
            self.beta = 10
        self._init_fused_state()

    def _init_fused_state(self):
        # state of the one-node forward pass (_DecodersFusedFn): never pickled, rebuilt on demand
        self.fused = True               # False: always the two-module path (encoder and decoder as separate autograd nodes)
        self.grid_bwd_flags = 0         # e.g. L.US_GRID_BWD_DETERMINISTIC: no float atomics in the table gradient
        self._fused_gen = 0
        self._fused_ws = None           # (tensor, key) scratch of the joint table gradient, grown on demand
        self._fused_ok = {}             # (grid geometry, n) -> bytes of that scratch (0: the joint kernels do not take the pair)
        self._mlp_ws = None
        self._flat = None               # packed [sdf decoder | colour decoder] vector that owns the nn.Linear parameters' storage
        self._layout = None
        self._pair = None

    # descriptors are ctypes objects: rebuild them after pickling / deepcopy (Tracker.py:106, UNISLAM.py:295-298)
    _TRANSIENT = ("_desc_sdf", "_desc_rgb", "_fused_ws", "_fused_ok", "_mlp_ws", "_flat", "_layout", "_pair")

    def __getstate__(self):
        s = self.__dict__.copy()
        for k in self._TRANSIENT:
            s.pop(k, None)
        return s

    def __setstate__(self, s):
        self.__dict__.update(s)
        self.__dict__.setdefault("mlp_precision", "fp32")
        fused, flags = self.__dict__.get("fused", True), self.__dict__.get("grid_bwd_flags", 0)
        self._init_fused_state()
        self.fused, self.grid_bwd_flags = fused, flags
        if not self.tcnn_network:
            self._desc_sdf, self._desc_rgb = self.mlp_descs()

    def mlp_descs(self):
        """(sdf, colour) us_mlp_desc of the two decoders"""
        if self.tcnn_network:
            return self.sdf_decoder.desc, self.color_decoder.desc
        if "_desc_sdf" not in self.__dict__:
            self._desc_sdf = make_mlp_desc(self.c_dim, self.hidden_size, self.n_blocks, 1, "tanh", True, self.mlp_precision)
            self._desc_rgb = make_mlp_desc(self.c_dim, self.hidden_size, self.n_blocks, 3, "sigmoid", True, self.mlp_precision)
        return self._desc_sdf, self._desc_rgb

    # ---- the one-node forward pass: packed parameters, cached scratch -------------------------------------------------------
    def _pack_layout(self):
        """[(parameter, offset)] of the decoders' parameters in the packed vector [sdf decoder | colour decoder], each decoder in
        us_mlp_desc order (weights, last matrix padded to 16 rows; then biases, last padded to 16) -- the order of _dec_params()"""
        if self._layout is None:
            d_s, d_c = self.mlp_descs()
            n_s, n_c = mlp_n_params(d_s), mlp_n_params(d_c)
            lay = []
            if self.tcnn_network:
                lay = [(self.sdf_decoder.params, 0), (self.color_decoder.params, n_s)]
            else:
                for base, hidden, out in ((0, self.linears, self.output_linear), (n_s, self.c_linears, self.c_output_linear)):
                    o = base
                    for l in hidden:
                        lay.append((l.weight, o)); o += l.weight.numel()
                    lay.append((out.weight, o)); o += 16 * out.weight.shape[1]
                    for l in hidden:
                        lay.append((l.bias, o)); o += l.bias.numel()
                    lay.append((out.bias, o)); o += 16
            self._layout, self._n_packed = lay, (n_s, n_c)
        return self._layout

    def _dec_params(self):
        return [p for p, _ in self._pack_layout()]

    def _packed_params(self):
        """device pointers of the two decoders' flat parameter vectors.  The nn.Linear parameters are kept as VIEWS of one packed
        buffer (their .data is re-pointed once; torch.optim.Adam, load_state_dict and state_dict work on the views), so nothing is
        concatenated per call; a buffer someone else laid out the same way (MapStep's flat parameter buffer) is used as it is, and
        after .to(device) / a re-pointing by anyone the parameters are packed again."""
        lay = self._pack_layout()
        n_s, n_c = self._n_packed
        base = lay[0][0].data_ptr()
        if not all(p.data_ptr() == base + 4 * o for p, o in lay) or (base & 15):
            p0 = lay[0][0]
            if not p0.is_cuda:
                raise L.UniSlamHipError("Decoders: the model must be on the GPU (unislam_amd has no CPU path)")
            flat = torch.zeros(n_s + n_c, dtype=torch.float32, device=p0.device)
            for p, o in lay:
                view = flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.detach())
                p.data = view
            self._flat = flat
            base = flat.data_ptr()
        return ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * n_s)

    def _pair_ok(self):
        if self._pair is None:
            d_s, d_c = self.mlp_descs()
            self._pair = bool(L.lib().us_mlp_pair_supported(ctypes.byref(d_s), ctypes.byref(d_c)))
        return self._pair

    def _fused_bytes(self, es, ec, n):
        """bytes of the joint table gradient's scratch for n points (0: the pair of grids / the batch is not taken by the joint kernels)"""
        a, b = es.desc, ec.desc
        key = (a.n_levels, a.log2_hashmap_size, b.log2_hashmap_size, a.base_resolution, a.per_level_scale, b.base_resolution,
               b.per_level_scale, a.n_features, b.n_features, n)
        v = self._fused_ok.get(key)
        if v is None:
            lib = L.lib()
            v = int(lib.us_hashgrid_joint_workspace_bytes(ctypes.byref(a), ctypes.byref(b), n)) \
                if lib.us_hashgrid_joint_supported(ctypes.byref(a), ctypes.byref(b), n) else 0
            if len(self._fused_ok) > 64:
                self._fused_ok.clear()
            self._fused_ok[key] = v
        return v

    def _fused_workspace(self, es, ec, n):
        nbytes = self._fused_bytes(es, ec, n)
        dev = es.params.device
        if self._fused_ws is None or self._fused_ws.numel() < nbytes or self._fused_ws.device != dev:
            self._fused_ws = None       # (free the old one first)
            self._fused_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._fused_gen += 1        # whatever a pending backward pass expects to find is gone
        return self._fused_ws, nbytes

    def _mlp_workspaces(self, dev):
        d_s, d_c = self.mlp_descs()
        lib = L.lib()
        nb = max(int(lib.us_mlp_bwd_workspace_bytes(ctypes.byref(d_s))), int(lib.us_mlp_bwd_workspace_bytes(ctypes.byref(d_c))))
        if self._mlp_ws is None or self._mlp_ws[0].device != dev or self._mlp_ws[0].numel() < nb:
            self._mlp_ws = (torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev))
        return self._mlp_ws[0], self._mlp_ws[1], nb

    def _fusable(self, es, ec, p_nor):
        return (self.fused and isinstance(es, HashGridEncoding) and isinstance(ec, HashGridEncoding) and es is not ec
                and p_nor.is_cuda and p_nor.shape[0] > 0 and es.params.is_cuda and ec.params.is_cuda
                and self._fused_bytes(es, ec, p_nor.shape[0]) > 0)

    def __deepcopy__(self, memo):
        new = Decoders(self.cfg, self.c_dim, self.hidden_size, self.truncation, self.n_blocks,
                       isinstance(self.beta, nn.Parameter))
        new.to(next(self.parameters()).device)
        new.load_state_dict(self.state_dict())
        for (_, a), (_, b) in zip(self.named_parameters(), new.named_parameters()):
            b.requires_grad_(a.requires_grad)
        if hasattr(self, "bound"):
            new.bound = self.bound
        new.fused, new.grid_bwd_flags = self.fused, self.grid_bwd_flags
        return new

    @staticmethod
    def pack_linear_params(hidden, out):
        """flat fp32 vector in us_mlp_desc layout: weights (last matrix zero-padded to 16 rows), then biases"""
        w = [l.weight.reshape(-1) for l in hidden]
        pad_w = out.weight.new_zeros((16 - out.weight.shape[0], out.weight.shape[1]))
        w.append(torch.cat([out.weight, pad_w], 0).reshape(-1))
        b = [l.bias for l in hidden]
        b.append(torch.cat([out.bias, out.bias.new_zeros(16 - out.bias.shape[0])]))
        return torch.cat(w + b)

    def sample_hash_grid_feature(self, p_nor, hash_grids_xyz):
        """decoders.py:91-105: clamp to [0,1], then encode (the clamp is folded into the HIP kernel)."""
        enc = hash_grids_xyz[0]
        if isinstance(enc, HashGridEncoding):
            return enc(p_nor, clamp=True)
        return enc(torch.clamp(p_nor, min=0, max=1))

    def get_raw_sdf(self, p_nor, scene_rep):
        """decoders.py:107-130"""
        hash_grids_xyz, c_hash_grids_xyz = scene_rep[0], scene_rep[1]
        h = self.sample_hash_grid_feature(p_nor, hash_grids_xyz)
        if self.tcnn_network:
            return self.sdf_decoder(h).squeeze()
        return fused_mlp(h, self.pack_linear_params(self.linears, self.output_linear), self.mlp_descs()[0]).squeeze()

    def get_raw_rgb(self, p_nor, scene_rep):
        """decoders.py:132-155"""
        hash_grids_xyz, c_hash_grids_xyz = scene_rep[0], scene_rep[1]
        h = self.sample_hash_grid_feature(p_nor, c_hash_grids_xyz)
        if self.tcnn_network:
            return self.color_decoder(h)
        return fused_mlp(h, self.pack_linear_params(self.c_linears, self.c_output_linear), self.mlp_descs()[1])

    def forward(self, p, scene_rep):
        """decoders.py:182-205: p [..., 3] (already normalised to [0,1]) -> raw [..., 4] = (rgb, sdf)"""
        p_shape = p.shape
        p_nor = p.reshape(-1, 3)
        es, ec = scene_rep[0][0], scene_rep[1][0]
        if self._fusable(es, ec, p_nor):
            raw = _DecodersFusedFn.apply(self, es, ec, torch.is_grad_enabled(), p_nor, es.params, ec.params, *self._dec_params())
            return raw.reshape(*p_shape[:-1], 4)
        sdf = self.get_raw_sdf(p_nor, scene_rep)
        rgb = self.get_raw_rgb(p_nor, scene_rep)
        raw = torch.cat([rgb, sdf.reshape(-1, 1)], dim=-1)
        return raw.reshape(*p_shape[:-1], -1)


def get_model(cfg):
    """reference src/networks/config.py:20-27"""
    return Decoders(cfg, c_dim=cfg['model']['c_dim'], truncation=cfg['model']['truncation'],
                    learnable_beta=cfg['rendering']['learnable_beta'])
