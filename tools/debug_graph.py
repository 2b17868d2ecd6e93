import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import unislam_amd as us
from test_gpu_step import _scene, BOUND, DEV
dec, es, ec = _scene(us, False, seed=7)
for p in dec.parameters(): p.requires_grad_(False)
H, Wd, fx, fy, cx, cy = 60, 80, 40.0, 40.0, 39.5, 29.5
g = torch.Generator().manual_seed(2)
gt_depth = (torch.rand(1, H, Wd, generator=g) * 1.5 + 0.5).to(DEV); gt_color = torch.rand(1, H, Wd, 3, generator=g).to(DEV)
n = 256
idx = torch.randint((H - 8) * (Wd - 8), (n,), generator=g).to(DEV); t_rand = torch.rand(n, 40, generator=g).to(DEV)
w = dict(fs=10, center=200, tail=50, color=5, depth=1)
for name, captured, capt_flag in (("eager capturable=False", False, False), ("eager capturable=True", False, True), ("graph", True, True)):
    pose = torch.nn.Parameter(torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]], device=DEV))
    opt = torch.optim.Adam([pose], lr=1e-3, betas=(0.5, 0.999), capturable=capt_flag)
    ts = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
    fn = lambda: ts.iterate(pose, gt_color, gt_depth, n, opt, H, Wd, fx, fy, cx, cy, 4, 4, t_rand=t_rand, indices=idx)
    it = us.CapturedIteration(fn, warmup=0) if captured else None
    for k in range(3):
        loss, _, _ = it.replay() if captured else fn()
        torch.cuda.synchronize()
        print(name, k, "loss", float(loss), "pose", [round(v, 5) for v in pose.detach().cpu().flatten().tolist()], "grad", [round(v, 3) for v in pose.grad.cpu().flatten().tolist()])
