import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R,"tests"), os.path.join(R,"oracle")): sys.path.insert(0,p)
import torch, numpy as np
import unislam_amd as us
from test_gpu_step import _scene, BOUND, DEV
dec, es, ec = _scene(us, False, seed=9)
for p in dec.parameters(): p.requires_grad_(False)
H, Wd, fx, fy, cx, cy = 60, 80, 40.0, 40.0, 39.5, 29.5
g = torch.Generator().manual_seed(4)
gt_depth = (torch.rand(1, H, Wd, generator=g) * 1.5 + 0.5).to(DEV); gt_depth[0, 20, 20:30] = 0.0
gt_color = torch.rand(1, H, Wd, 3, generator=g).to(DEV)
n, eh, ew = 300, 4, 5
w = dict(fs=10, center=200, tail=50, color=5, depth=1)
pose0 = torch.tensor([[0.9, 0.1, -0.2, 0.3, 3.0, 1.2, 0.0]], device=DEV)
idx = torch.randint((H - 2 * eh) * (Wd - 2 * ew), (n,), generator=g).to(DEV); tr = torch.rand(n, 40, generator=g).to(DEV)
quad = torch.nn.Parameter(pose0[:, :4].clone()); T = torch.nn.Parameter(pose0[:, 4:].clone())
opt = torch.optim.SGD([quad, T], lr=0.0)
ts_a = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
ts_b = us.TrackStep(es, ec, dec, BOUND, 32, 8, 0.06, w, max_rays=n)
ts_b.begin_frame(pose0, gt_color[0], gt_depth[0], 0.0, 0.0, H, Wd, fx, fy, cx, cy, eh, ew)
la, ua, va = ts_a.iterate(torch.cat([quad, T], -1), gt_color, gt_depth, n, opt, H, Wd, fx, fy, cx, cy, eh, ew, t_rand=tr, indices=idx)
lb, ub, vb = ts_b.iterate_fused(n, t_rand=tr, indices=idx)
torch.cuda.synchronize()
print("loss", float(la), float(lb))
print("g_o equal", torch.allclose(ts_a.g_o, ts_b.g_o, rtol=1e-5, atol=1e-7), (ts_a.g_o-ts_b.g_o).abs().max().item(), ts_a.g_o.abs().max().item())
print("g_d equal", torch.allclose(ts_a.g_d, ts_b.g_d, rtol=1e-5, atol=1e-7), (ts_a.g_d-ts_b.g_d).abs().max().item())
print("sum g_o a", ts_a.g_o.sum(0).tolist(), "T.grad", T.grad.tolist(), "fused", ts_b.g_pose[4:].tolist())
print("quad.grad", quad.grad.tolist(), "fused", ts_b.g_pose[:4].tolist())
print("rays equal", (ts_b.t_rd - us.common.get_rays_from_uv((ew + idx % (Wd-2*ew)).float()[None], (eh + idx // (Wd-2*ew)).float()[None], us.common.cam_pose_to_matrix(pose0), H, Wd, fx, fy, cx, cy, DEV)[1].reshape(-1,3)).abs().max().item())
