"""
Adam -- drop-in for the `torch.optim.Adam(params_list)` of reference src/Mapper.py:111-139,364 (and src/Tracker.py:328): same
constructor (an iterable of parameters or of param-group dicts with their own `lr`), same `step()` / `zero_grad()` / `state_dict()`,
same arithmetic per element (the op order of torch's single-tensor Adam: lerp, mul + addcmul, sqrt / div / add, addcdiv; no
weight decay, no amsgrad -- the reference uses neither), but the whole `optimizer.step()` is ONE launch over all parameters of all
groups (us_adam_step_tensors) instead of ~10 foreach passes per group: at room0's sizes 7 streams x 51.7 MB instead of ~25.

    - optimizer = torch.optim.Adam([{'params': decoders_para_list, 'lr': 0}, {'params': hash_grids_para, 'lr': 0}, ...])
    + optimizer = unislam_amd.optim.Adam([...the same list...])

The reference sets the learning rates through `optimizer.param_groups[k]['lr']` afterwards (src/Mapper.py:123-126): read per step here too.
"""
import ctypes

import torch

from . import _lib as L

_MAX = 40          # ADAM_MAX_TENSORS of csrc/render.hip


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise ValueError("unislam_amd.optim.Adam: weight_decay / amsgrad are not part of Uni-SLAM's optimisers")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # one launch per (betas, eps, step count) class -- in Uni-SLAM: one launch
        batches = {}
        for group in self.param_groups:
            lr, (b1, b2), eps = float(group["lr"]), group["betas"], float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.UniSlamHipError("unislam_amd.optim.Adam: contiguous fp32 GPU parameters only (there is no CPU path)")
                g = p.grad
                if g.is_sparse:
                    raise L.UniSlamHipError("unislam_amd.optim.Adam: dense gradients only")
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                batches.setdefault((float(b1), float(b2), eps, st["step"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"], lr))
        lib, stream = L.lib(), None
        for (b1, b2, eps, step), items in batches.items():
            stream = L.stream() if stream is None else stream
            for i in range(0, len(items), _MAX):
                part = items[i:i + _MAX]
                k = len(part)
                VP, I64, DBL = ctypes.c_void_p * k, ctypes.c_int64 * k, ctypes.c_double * k
                L.check(lib.us_adam_step_tensors(k, VP(*[t[0].data_ptr() for t in part]), VP(*[t[1].data_ptr() for t in part]),
                                                 VP(*[t[2].data_ptr() for t in part]), VP(*[t[3].data_ptr() for t in part]),
                                                 I64(*[t[0].numel() for t in part]), DBL(*[t[4] for t in part]), b1, b2, eps, step, stream),
                        "us_adam_step_tensors")
                for t in part:
                    # the kernel wrote the parameter through its raw pointer: tell autograd (a backward pass that saved it, or the decoders'
                    # one-node forward pass that recorded its version, must see the in-place update as torch.optim.Adam's would be seen)
                    torch.autograd.graph.increment_version(t[0])
        return loss
