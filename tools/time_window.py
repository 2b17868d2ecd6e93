"""time the joint_opt mapping iteration (window.MapWindow) at the bench scene: eager and hipGraph replay; development tool
   python tools/time_window.py [b n_per extra_frames extra_n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
bench.torch = torch
import unislam_amd as us
dev = "cuda:0"
bound = bench.load_bound(bench.ROOM0_BOUND)
pls = bench.per_level_scale(816)
b, n_per, xf, xn = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (16, 256, 0, 0))]
joint = int(os.environ.get("JOINT_OPT", "1"))


def build():
    torch.manual_seed(0)
    dec = us.Decoders({"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}, c_dim=32, hidden_size=32,
                      truncation=0.06, n_blocks=2).to(dev)
    mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                            "base_resolution": 16, "per_level_scale": pls}).to(dev)
    es, ec = mk(16), mk(19)
    R = b * n_per + min(xf, b) * xn
    step = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, bench.W, bench.LR, max_rays=R, deterministic=bool(int(os.environ.get("DET", "0"))))
    c2ws, pd, pc, pr = bench.keyframe_pools(b, bound, 1000, dev)
    win = us.MapWindow(step, c2ws, pd, pc, pr, n_per, joint_opt=bool(joint), cam_lr=1e-3, extra=(xf, xn) if xf else None, has_zero_depth=False)
    return step, win


def timed(fn, k=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


step, win = build()
e = timed(win.iterate)
U = int(os.environ.get("UNROLL", "1"))
win.capture(unroll=U)
g = timed(win.replay, k=200 // U) / U
print(f"window b={b} n_per={n_per} extra={xf}x{xn} rays={win.R} joint_opt={joint}: eager {e:.4f} ms, graph {g:.4f} ms, "
      f"{win.R / g / 1e3:.2f} M rays/s; loss {float(win.replay()):.5f}")
if os.environ.get("PROBE"):
    step.probe, step.probe_every, step._it = {}, 1, 0
    for _ in range(10):
        win.iterate()
    torch.cuda.synchronize()
    for k, v in sorted(step.probe.items()):
        print(f"  {k:28s} {sum(a.elapsed_time(c) for a, c in v) / len(v) * 1e3:8.1f} us")
