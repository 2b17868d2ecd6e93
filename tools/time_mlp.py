import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unislam_amd as us
from unislam_amd import _lib as L
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for width, nh, prec in [(w, h, pr) for (w, h) in [(32, 2), (16, 2), (16, 1), (64, 2)] for pr in ("fp32", "bf16")]:
    desc = us.make_mlp_desc(32, width, nh, 3, "sigmoid", True, prec)
    p = torch.randn(us.network.mlp_n_params(desc), device=DEV) * 0.3
    x = torch.randn(n, 32, device=DEV); y = torch.empty(n, 3, device=DEV); dy = torch.randn(n, 3, device=DEV)
    dx = torch.empty(n, 32, device=DEV); gp = torch.zeros_like(p)
    lib = L.lib(); st = L.stream()
    wsb = int(lib.us_mlp_bwd_workspace_bytes(ctypes.byref(desc))); ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    L.check(lib.us_mlp_fwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), n, L.ptr(y), 3, 0, st), "f")
    def t(fn, reps=20):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    full = t(lambda: lib.us_mlp_bwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), L.ptr(y), 3, L.ptr(dy), 3, n, L.ptr(dx), L.ptr(gp), 0, L.ptr(ws), wsb, st))
    only_dx = t(lambda: lib.us_mlp_bwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), L.ptr(y), 3, L.ptr(dy), 3, n, L.ptr(dx), None, 0, None, 0, st))
    only_gp = t(lambda: lib.us_mlp_bwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), L.ptr(y), 3, L.ptr(dy), 3, n, None, L.ptr(gp), 0, L.ptr(ws), wsb, st))
    fwd = t(lambda: lib.us_mlp_fwd(ctypes.byref(desc), L.ptr(p), L.ptr(x), n, L.ptr(y), 3, 0, st))
    print(f"{prec} width {width} hidden {nh}: fwd {fwd:.1f} us  bwd full {full:.1f}  only dL_din {only_dx:.1f}  only grad_params {only_gp:.1f}")
