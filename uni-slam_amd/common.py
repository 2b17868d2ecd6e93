"""
Ray generation / pixel sampling helpers with the API of reference src/common.py (same names, argument order and
return values).  The mapping-time ray assembly (get_samples_all) runs the gather-then-rotate HIP kernel
(us_gather_rays) instead of rotating the whole [b,P,3] pool and gathering afterwards (common.py:160-164).
"""
import ctypes

import numpy as np
import torch

from . import _lib as L


def as_intrinsics_matrix(intrinsics):
    """common.py:22-33"""
    K = np.eye(3)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]
    return K


def _dirs(i, j, fx, fy, cx, cy):
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def get_camera_rays(H, W, fx, fy=None, cx=None, cy=None, type='OpenGL'):
    """common.py:35-46: camera-frame directions [H,W,3] (OpenGL: x right, y up, -z forward)."""
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing='xy')
    if type == 'OpenGL':
        return _dirs(i, j, fx, fy, cx, cy)
    if type == 'OpenCV':
        return torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], -1)
    raise ValueError(type)


def sample_pdf(bins, weights, N_samples, det=False, device='cuda:0'):
    """common.py:49-85 (keeps the reference's un-normalised pdf, :55-56)."""
    cdf = torch.cumsum(weights, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0., 1., steps=N_samples, device=device).expand(list(cdf.shape[:-1]) + [N_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [N_samples], device=device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def random_select(l, k):
    """common.py:88-93"""
    return list(np.random.permutation(np.array(range(l)))[:min(l, k)])


def get_rays_from_uv(i, j, c2ws, H, W, fx, fy, cx, cy, device):
    """common.py:95-107 (differentiable wrt c2ws: the tracker's pose gradient flows through here)."""
    dirs = _dirs(i, j, fx, fy, cx, cy).unsqueeze(-2)
    rays_d = torch.sum(dirs * c2ws[:, None, :3, :3], -1)
    rays_o = c2ws[:, None, :3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def select_uv(i, j, n, b, depths, colors, device='cuda:0'):
    """common.py:109-131"""
    i, j = i.reshape(-1), j.reshape(-1)
    indices = torch.randint(i.shape[0], (n * b,), device=device)
    indices = indices.clamp(0, i.shape[0])
    i, j = i[indices], j[indices]
    indices = indices.reshape(b, -1)
    i, j = i.reshape(b, -1), j.reshape(b, -1)
    depths = depths.reshape(b, -1)
    colors = colors.reshape(b, -1, 3)
    depths = torch.gather(depths, 1, indices)
    colors = torch.gather(colors, 1, indices.unsqueeze(-1).expand(-1, -1, 3))
    return i, j, depths, colors


def get_sample_uv(H0, H1, W0, W1, n, b, depths, colors, device='cuda:0'):
    """common.py:133-150"""
    if not (H0 == 0 and W0 == 0):
        depths = depths[:, H0:H1, W0:W1]
        colors = colors[:, H0:H1, W0:W1]
    i, j = torch.meshgrid(torch.linspace(W0, W1 - 1, W1 - W0, device=device),
                          torch.linspace(H0, H1 - 1, H1 - H0, device=device), indexing='ij')
    return select_uv(i.t(), j.t(), n, b, depths, colors, device=device)


def get_samples_all(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors, device, rays_d, indices=None, out=None):
    """
    common.py:152-166.  depths [b,P], colors [b,P,3], rays_d [b,P,3] are per-frame pixel pools.
    `indices` ([b,n] int64, optional) replaces the torch.randint draw of :155 (parity tests).
    `out` (optional): (rays_o [b*n,3], rays_d [b*n,3], depth [b*n], color [b*n,3]) fp32 tensors the kernel writes into -- the
    static inputs of a captured iteration (MapStep.capture) are refilled this way without a copy.
    """
    b, P = depths.shape
    if indices is None:
        indices = torch.randint(P, (n * b,), device=device).reshape(b, -1)
    if not depths.is_cuda:
        raise L.UniSlamHipError("get_samples_all: the pixel pools must be on the GPU (unislam_amd has no CPU path)")
    if c2ws.requires_grad:
        # joint pose optimisation (Mapper.py:372-376) through autograd: keep the rotation in the graph (torch ops on the GPU).
        # (window.MapWindow is the kernel path for this case: poses on the device, no autograd)
        if out is not None:
            raise L.UniSlamHipError("get_samples_all: `out` cannot be combined with c2ws.requires_grad (the autograd branch returns "
                                    "fresh tensors; a captured iteration would replay on stale rays)")
        sd = torch.gather(depths, 1, indices)
        sc = torch.gather(colors, 1, indices.unsqueeze(-1).expand(-1, -1, 3))
        gi = indices.unsqueeze(-1).expand(-1, -1, 3)
        d_sel = torch.gather(rays_d, 1, gi)
        rd = torch.sum(d_sel.unsqueeze(-2) * c2ws[:, None, :3, :3], -1)
        ro = c2ws[:, None, :3, -1].expand(rd.shape)
        return ro.reshape(-1, 3), rd.reshape(-1, 3), sd.reshape(-1), sc.reshape(-1, 3)
    n_per = indices.shape[1]
    tot = b * n_per
    if out is not None:
        ro, rd, sd, sc = out
        if not (ro.shape == (tot, 3) and rd.shape == (tot, 3) and sd.shape == (tot,) and sc.shape == (tot, 3)
                and all(t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda for t in out)):
            raise L.UniSlamHipError(f"get_samples_all: `out` must be contiguous fp32 GPU tensors of {tot} rays")
    else:
        ro = torch.empty((tot, 3), dtype=torch.float32, device=depths.device)
        rd = torch.empty_like(ro); sc = torch.empty_like(ro)
        sd = torch.empty((tot,), dtype=torch.float32, device=depths.device)
    c = L.f32(c2ws.detach()); pd = L.f32(depths); pc = L.f32(colors); pr = L.f32(rays_d); ix = indices.contiguous()
    L.check(L.lib().us_gather_rays(L.ptr(c), L.ptr(pd), L.ptr(pc), L.ptr(pr), L.ptr(ix), b, P, n_per, L.ptr(ro),
                                   L.ptr(rd), L.ptr(sd), L.ptr(sc), L.stream()), "us_gather_rays")
    return ro, rd, sd, sc


def get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws, depths, colors, device):
    """common.py:168-180"""
    b = c2ws.shape[0]
    i, j, sample_depth, sample_color = get_sample_uv(H0, H1, W0, W1, n, b, depths, colors, device=device)
    rays_o, rays_d = get_rays_from_uv(i, j, c2ws, H, W, fx, fy, cx, cy, device)
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), sample_depth.reshape(-1), sample_color.reshape(-1, 3)


# ---- pose helpers (common.py:182-208; pytorch3d.transforms restated, real-first quaternions) ----
def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    m = x > 0
    ret[m] = torch.sqrt(x[m])
    return ret


def matrix_to_quaternion(matrix):
    """pytorch3d.transforms.matrix_to_quaternion (real part first, numerically safe branch selection)."""
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    best = torch.nn.functional.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return quat_candidates[best, :].reshape(batch_dim + (4,))


def _on_device_plain(t):
    return t.is_cuda and t.dtype == torch.float32 and not t.requires_grad and t.dim() == 3


def matrix_to_cam_pose(batch_matrices, RT=True):
    """common.py:182-194.  Device tensors outside autograd take one launch (us_matrix_to_cam_pose) instead of ~30 torch ops."""
    if RT and _on_device_plain(batch_matrices) and tuple(batch_matrices.shape[1:]) == (4, 4):
        m = batch_matrices.contiguous()
        out = torch.empty((m.shape[0], 7), dtype=torch.float32, device=m.device)
        L.check(L.lib().us_matrix_to_cam_pose(L.ptr(m), m.shape[0], 0, L.ptr(out), L.stream()), "us_matrix_to_cam_pose")
        return out
    if RT:
        return torch.cat([matrix_to_quaternion(batch_matrices[:, :3, :3]), batch_matrices[:, :3, 3]], dim=-1)
    return torch.cat([batch_matrices[:, :3, 3], matrix_to_quaternion(batch_matrices[:, :3, :3])], dim=-1)


def predict_cam_pose(c2w_before, c2w_last):
    """the tracker's constant-speed initial guess (Tracker.py:317-320): 2 * pose(c2w_last) - pose(c2w_before) on the 7 numbers -> [1, 7]"""
    m = torch.stack([c2w_before, c2w_last], dim=0)
    if _on_device_plain(m):
        out = torch.empty((1, 7), dtype=torch.float32, device=m.device)
        L.check(L.lib().us_matrix_to_cam_pose(L.ptr(m), 1, 1, L.ptr(out), L.stream()), "us_matrix_to_cam_pose")
        return out
    pre = matrix_to_cam_pose(m)
    if float((pre[0, :4] * pre[1, :4]).sum()) < 0:           # q and -q are one rotation: extrapolate on one hemisphere (csrc/window.hip)
        pre = torch.cat([pre[:1], torch.cat([-pre[1:, :4], pre[1:, 4:]], -1)], 0)
    return 2 * pre[1:] - pre[0:1]


def cam_pose_to_matrix(batch_poses):
    """common.py:196-208.  Device tensors outside autograd take one launch (us_cam_pose_to_matrix)."""
    if batch_poses.is_cuda and batch_poses.dtype == torch.float32 and not batch_poses.requires_grad and batch_poses.dim() == 2 \
            and batch_poses.shape[1] == 7:
        p = batch_poses.contiguous()
        out = torch.empty((p.shape[0], 4, 4), dtype=torch.float32, device=p.device)
        L.check(L.lib().us_cam_pose_to_matrix(L.ptr(p), p.shape[0], L.ptr(out), L.stream()), "us_cam_pose_to_matrix")
        return out
    c2w = torch.eye(4, device=batch_poses.device).unsqueeze(0).repeat(batch_poses.shape[0], 1, 1)
    c2w[:, :3, :3] = quaternion_to_matrix(batch_poses[:, :4])
    c2w[:, :3, 3] = batch_poses[:, 4:]
    return c2w


def get_rays(H, W, fx, fy, cx, cy, c2w, device):
    """common.py:210-228: rays of a whole image, [H,W,3] each."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing='ij')
    i, j = i.t(), j.t()
    dirs = _dirs(i, j, fx, fy, cx, cy).to(device).reshape(H, W, 1, 3)
    c2w = c2w.to(device)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def normalize_3d_coordinate(p, bound):
    """common.py:231-245: to [-1,1] (in place on the reshaped view, like the reference)."""
    p = p.reshape(-1, 3)
    p[:, 0] = ((p[:, 0] - bound[0, 0]) / (bound[0, 1] - bound[0, 0])) * 2 - 1.0
    p[:, 1] = ((p[:, 1] - bound[1, 0]) / (bound[1, 1] - bound[1, 0])) * 2 - 1.0
    p[:, 2] = ((p[:, 2] - bound[2, 0]) / (bound[2, 1] - bound[2, 0])) * 2 - 1.0
    return p


def bbox_filter(rays_o, rays_d, gt_depth, bound, require_depth=False):
    """
    Mapper.py:396-402 / Tracker.py:177-184: bool mask `far_bb >= gt_depth` (& gt_depth > 0 for tracking),
    one kernel, no [R,3,2] temporaries.  bound: CPU or GPU tensor [3,2].
    """
    o, d, g = L.f32(rays_o.detach()), L.f32(rays_d.detach()), L.f32(gt_depth.detach())
    valid = torch.empty(o.shape[0], dtype=torch.uint8, device=o.device)
    L.check(L.lib().us_bbox_filter(L.ptr(o), L.ptr(d), L.ptr(g), bound_host(bound), o.shape[0], int(require_depth),
                                   L.ptr(valid), None, L.stream()), "us_bbox_filter")
    return valid.view(torch.bool)           # (the kernel writes 0 / 1: one byte per ray either way)


def bbox_far(rays_o, rays_d, bound):
    """Renderer.py:108-111: far intersection with the scene box, [R]."""
    o, d = L.f32(rays_o.detach()), L.f32(rays_d.detach())
    far = torch.empty(o.shape[0], dtype=torch.float32, device=o.device)
    L.check(L.lib().us_bbox_filter(L.ptr(o), L.ptr(d), None, bound_host(bound), o.shape[0], 0, None, L.ptr(far),
                                   L.stream()), "us_bbox_filter")
    return far


def bound_host(bound):
    """
    [3,2] tensor -> host float[6] = lo[3], hi[3].  No cache: a cache keyed on data_ptr() handed a later scene's bound tensor,
    allocated at the same address, the old scene's values.  A GPU tensor costs one synchronising copy per call, so the
    long-lived callers (MapStep, TrackStep, Renderer) convert once in __init__ and pass the host array on; an array made by
    this function passes through unchanged.
    """
    if isinstance(bound, ctypes.Array):
        return bound
    b = bound.detach().float().cpu()
    return L.host_floats([b[0, 0], b[1, 0], b[2, 0], b[0, 1], b[1, 1], b[2, 1]])
