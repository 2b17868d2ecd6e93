"""Forward deviation of the bf16-operand decoders against the fp32 ones on IDENTICAL parameters (development tool)."""
import sys, os
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests")); sys.path.insert(0, os.path.join(R0, "oracle"))
import torch, unislam_amd as us
import test_gpu_step as T
DEV = "cuda:0"
R, S = 4096, 64
ro, rd, gd, gc = T._rays(R, seed=5)
t_rand = torch.rand(R, S, device=DEV)
for tcnn, hidden in ((False, 32), (True, 16)):
    steps = {}
    for prec in ("fp32", "bf16", "bf16_plain", "f16"):
        torch.manual_seed(3)
        cfg = dict(T._cfg(tcnn, 48, 16), model={"mlp_precision": prec})
        dec = us.Decoders(cfg, c_dim=32, hidden_size=hidden, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, T._ecfg(16)).to(DEV), us.HashGridEncoding(3, T._ecfg(19)).to(DEV)
        steps[prec] = us.MapStep(es, ec, dec, T.BOUND, 48, 16, 0.06, T.W, T.LR, max_rays=R)
    for _ in range(100):                                         # train in fp32 so the field is not noise
        steps["fp32"].iterate(ro, rd, gd, gc, has_zero_depth=False)
    for k in ("bf16", "bf16_plain", "f16"):
        steps[k].flat.copy_(steps["fp32"].flat)
    outs = {}
    for prec in ("fp32", "bf16", "bf16_plain", "f16"):
        steps[prec].forward(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)
        outs[prec] = [t.clone() for t in steps[prec].rendered()[:4]]
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    med = lambda a, b: ((a - b).abs() / b.abs().clamp(min=1e-3)).median().item()
    for prec in ("bf16", "bf16_plain", "f16"):
        mx = lambda a, b: ((a - b).abs() / b.abs().clamp(min=1e-3)).max().item()
        print(prec, "tcnn", tcnn, "hidden", hidden, "| depth: norm-rel %.2e median-rel %.2e max-rel %.2e | rgb: norm-rel %.2e median-rel %.2e max-rel %.2e" %
              (rel(outs[prec][2], outs["fp32"][2]), med(outs[prec][2], outs["fp32"][2]), mx(outs[prec][2], outs["fp32"][2]),
               rel(outs[prec][3], outs["fp32"][3]), med(outs[prec][3], outs["fp32"][3]), mx(outs[prec][3], outs["fp32"][3])))
