"""development: the two-decoder launches at the bench shape (262 144 points, 2 x 32, bf16 split, level-major planes, raw[N][4] outputs):
us_mlp_fwd_pair / us_mlp_bwd_pair (+ reduce) in us, back-to-back launches timed with HIP events.  python tools/time_mlp_pair.py [n] [precision] [split]
(split: the input planes taken as the hi / lo bf16 pairs of US_MLP_IN_SPLIT_BF16 -- the float planes' bits, reinterpreted: a timing run)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unislam_amd as us
from unislam_amd import _lib as L
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
SPL = L.US_MLP_IN_SPLIT_BF16 if (len(sys.argv) > 3 and sys.argv[3] == "split") else 0
lib, st = L.lib(), L.stream()
ds = us.make_mlp_desc(32, 32, 2, 1, "tanh", True, prec); dc = us.make_mlp_desc(32, 32, 2, 3, "sigmoid", True, prec)
ps = torch.randn(us.network.mlp_n_params(ds), device=DEV) * 0.3; pc = torch.randn(us.network.mlp_n_params(dc), device=DEV) * 0.3
fa, fb = torch.randn(16, n, 2, device=DEV), torch.randn(16, n, 2, device=DEV)
raw, d_raw = torch.empty(n, 4, device=DEV), torch.randn(n, 4, device=DEV)
da, db = torch.empty(16, n, 2, device=DEV), torch.empty(16, n, 2, device=DEV)
gs, gc = torch.zeros_like(ps), torch.zeros_like(pc)
wsb = int(lib.us_mlp_bwd_workspace_bytes(ctypes.byref(ds)))
wa, wb = torch.empty(wsb, dtype=torch.uint8, device=DEV), torch.empty(wsb, dtype=torch.uint8, device=DEV)
off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
P = L.ptr
A, B = ctypes.byref(ds), ctypes.byref(dc)
fwd = lambda: L.check(lib.us_mlp_fwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), n, off(raw, 3), 4, P(raw), 4, 1 | SPL, st), "fwd")
bwd = lambda: L.check(lib.us_mlp_bwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), off(raw, 3), 4, P(raw), 4, off(d_raw, 3), 4, P(d_raw), 4, n, P(da), P(db),
                                          P(gs), P(gc), 1 | SPL | L.US_MLP_DEFER_REDUCE, P(wa), P(wb), wsb, st), "bwd")
bwd_in = lambda: L.check(lib.us_mlp_bwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), off(raw, 3), 4, P(raw), 4, off(d_raw, 3), 4, P(d_raw), 4, n, P(da), P(db),
                                             None, None, 1 | SPL, None, None, 0, st), "bwd_in")
bwd_w = lambda: L.check(lib.us_mlp_bwd_pair(A, B, P(ps), P(pc), P(fa), P(fb), off(raw, 3), 4, P(raw), 4, off(d_raw, 3), 4, P(d_raw), 4, n, None, None,
                                            P(gs), P(gc), 1 | SPL | L.US_MLP_DEFER_REDUCE, P(wa), P(wb), wsb, st), "bwd_w")
def t(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
TAG = " (pre-split inputs)" if SPL else ""
print(f"{prec}{TAG} n {n}: fwd_pair {t(fwd):.1f} us  bwd_pair {t(bwd):.1f} us  bwd_pair (input gradients only) {t(bwd_in):.1f} us  "
      f"bwd_pair (parameter gradients only) {t(bwd_w):.1f} us")
