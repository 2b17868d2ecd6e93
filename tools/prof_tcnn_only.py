"""where the `drop_in_api.tcnn_only` iteration (bench.py _TcnnOnlyModel: the reference's own torch code around tcnn.Encoding / tcnn.Network)
spends its time: host issue time against wall time, then the GPU time by kernel (torch profiler), 20 iterations
   python tools/prof_tcnn_only.py [fused]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench.torch = torch
import unislam_amd as us
import unislam_amd.tcnn as tcnn

dev = "cuda:0"
bound = bench.load_bound(bench.ROOM0_BOUND)
c2ws, pd, pc, pr = bench.keyframe_pools(16, bound, 5000, dev)
torch.manual_seed(0)
m = bench._tcnn_only_model_class()(tcnn, 32, "bf16", (16, 19), 816, bound.to(dev), 0.06, bench.W, 48, 16).to(dev)
groups = [{"params": list(m.sdf_decoder.parameters()) + list(m.color_decoder.parameters()) + [m.beta], "lr": 1e-3},
          {"params": [m.enc_s.params], "lr": 0.05}, {"params": [m.enc_c.params], "lr": 0.05}]
opt = us.optim.Adam(groups) if "fused" in sys.argv[1:] else torch.optim.Adam(groups)
it = lambda: m.iteration(opt, c2ws, pd, pc, pr, 256)
for _ in range(20):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    it()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time {20 * t_host:.3f} ms/iter, wall {20 * t_all:.3f} ms/iter")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
    for _ in range(20):
        it()
    torch.cuda.synchronize()
rows = [(e.key, e.device_time_total / 20 / 1e3, e.count / 20) for e in p.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
rows.sort(key=lambda r: -r[1])
tot = sum(r[1] for r in rows)
print(f"GPU time by kernel, ms per iteration (sum {tot:.3f}):")
for k, t, c in rows[:40]:
    print(f"  {t:8.4f}  x{c:5.1f}  {k[:150]}")
print(f"  {sum(r[1] for r in rows[40:]):8.4f}  the other {len(rows) - 40} kernels, {sum(r[2] for r in rows):.0f} launches per iteration in all")
