"""
FusedMLP -- drop-in for tcnn.Network(otype="FullyFusedMLP") (reference src/networks/decoders.py:50-70) and the
compute engine behind the nn.Linear decoder stacks (decoders.py:74-84): one MFMA kernel per pass
(csrc/mlp.hip), weights resident in LDS.

    net = FusedMLP(n_input_dims=32, n_output_dims=1,
                   network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Tanh",
                                   "n_neurons": 16, "n_hidden_layers": 1})
    y = net(feat)            # [N, 32] -> [N, n_output_dims]
    net.params               # flat fp32: W0[width][32], hidden W[width][width] ..., Wlast[16][width]  (tcnn layout)
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib as L

ACT = {"none": 0, "None": 0, "tanh": 1, "Tanh": 1, "sigmoid": 2, "Sigmoid": 2}
# MFMA operand type (us_mlp_desc.precision): parameters, gradients and accumulation are fp32 in both
PREC = {"fp32": 0, "f32": 0, "float": 0, "bf16": 1, "bfloat16": 1, "bf16_plain": 2, "f16": 3, "fp16": 3, "half": 3}   # bf16: split-operand forward products (network.py docs)


def make_mlp_desc(n_in, width, n_hidden, n_out, out_act, has_bias, precision=0):
    d = L.MlpDesc(n_in, width, n_hidden, n_out, ACT[out_act] if isinstance(out_act, str) else int(out_act),
                  1 if has_bias else 0, PREC[precision] if isinstance(precision, str) else int(precision))
    return d


def mlp_n_params(desc):
    return int(L.lib().us_mlp_n_params(ctypes.byref(desc)))


class _MlpFn(torch.autograd.Function):
    """y[N, n_out] = MLP(x[N, 32]; params).  `out` may be a preallocated strided view target (raw[N,4] columns)."""

    @staticmethod
    def forward(ctx, x, params, desc, owner=None):
        x = L.f32(x.detach())
        p = L.f32(params.detach())
        n = x.shape[0]
        out = torch.empty((n, desc.n_out), dtype=torch.float32, device=x.device)
        L.check(L.lib().us_mlp_fwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), n, L.ptr(out), desc.n_out, 0, L.stream()),
                "us_mlp_fwd")
        ctx.desc, ctx.owner = desc, owner
        ctx.save_for_backward(x, p, out)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, p, out = ctx.saved_tensors
        desc = ctx.desc
        dy = L.f32(dy)
        n = x.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gp = torch.zeros_like(p) if ctx.needs_input_grad[1] else None
        if gx is not None or gp is not None:
            ws, nbytes = None, 0
            if gp is not None:
                nbytes = int(L.lib().us_mlp_bwd_workspace_bytes(ctypes.byref(desc)))
                if ctx.owner is not None:         # the partial-row scratch is cached on the module (consumed inside this one call)
                    if ctx.owner._ws is None or ctx.owner._ws.numel() < nbytes or ctx.owner._ws.device != x.device:
                        ctx.owner._ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                    ws = ctx.owner._ws
                else:
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            L.check(L.lib().us_mlp_bwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), L.ptr(out), desc.n_out, L.ptr(dy),
                                       desc.n_out, n, L.ptr(gx), L.ptr(gp), 0, L.ptr(ws), nbytes, L.stream()), "us_mlp_bwd")
        return gx, gp, None, None


def fused_mlp(x, params, desc):
    return _MlpFn.apply(x, params, desc)


class FusedMLP(nn.Module):
    def __init__(self, n_input_dims=32, n_output_dims=1, network_config=None, bias=False, seed=1337):
        super().__init__()
        c = dict(network_config or {})
        if c.get("activation", "ReLU") != "ReLU":
            raise ValueError("FusedMLP: hidden activation must be ReLU")
        self.network_config = c
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.width = int(c.get("n_neurons", 16))
        self.n_hidden = int(c.get("n_hidden_layers", 1))
        self.bias = bool(bias)
        self.desc = make_mlp_desc(n_input_dims, self.width, self.n_hidden, n_output_dims,
                                  c.get("output_activation", "None"), self.bias, c.get("precision", "fp32"))
        n = mlp_n_params(self.desc)
        g = torch.Generator().manual_seed(seed)
        # xavier-uniform per matrix like tcnn's FullyFusedMLP::initialize_params; biases (if any) start at zero
        p = torch.zeros(n)
        o = 0
        for (fo, fi) in self.layer_shapes():
            lim = math.sqrt(6.0 / (fi + fo))
            p[o:o + fo * fi] = (torch.rand(fo * fi, generator=g) * 2 - 1) * lim
            o += fo * fi
        self.params = nn.Parameter(p)
        self._ws = None                 # cached scratch of the backward pass (never pickled)

    def layer_shapes(self):
        s = [(self.width, self.n_input_dims)] + [(self.width, self.width)] * (self.n_hidden - 1) + [(16, self.width)]
        return s

    def __getstate__(self):
        s = self.__dict__.copy()
        s.pop("desc", None)
        s["_ws"] = None
        return s

    def __setstate__(self, s):
        self.__dict__.update(s)
        self.__dict__.setdefault("_ws", None)
        self.desc = make_mlp_desc(self.n_input_dims, self.width, self.n_hidden, self.n_output_dims,
                                  self.network_config.get("output_activation", "None"), self.bias,
                                  self.network_config.get("precision", "fp32"))

    def __deepcopy__(self, memo):
        new = FusedMLP(self.n_input_dims, self.n_output_dims, self.network_config, self.bias)
        new.params = nn.Parameter(self.params.detach().clone(), requires_grad=self.params.requires_grad)
        return new

    def forward(self, x):
        if x.dim() != 2 or x.shape[1] != self.n_input_dims:
            raise ValueError(f"FusedMLP: expected [N,{self.n_input_dims}], got {tuple(x.shape)}")
        return _MlpFn.apply(x, self.params, self.desc, self)
