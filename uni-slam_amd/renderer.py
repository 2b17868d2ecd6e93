"""
Renderer -- API of reference src/utils/Renderer.py (class Renderer, :21-223) on the HIP kernels of csrc/render.hip:
depth-guided z sampling + jitter (us_sample_z), points in the unit cube (us_ray_points), SDF->alpha compositing
with its five per-ray reductions (us_composite_fwd/bwd).  render_batch_ray keeps the reference's argument order
(rays_d BEFORE rays_o) and returns the same 7-tuple.
"""
import ctypes

import torch

from . import _lib as L
from .common import get_rays, sample_pdf, normalize_3d_coordinate, bbox_far, bound_host


class _RayPointsFn(torch.autograd.Function):
    """pts[R,S,3] = ((o + d z) - lo) / (hi - lo)   (Renderer.py:132-137); z carries no gradient."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, bhost):
        o, d, z = L.f32(rays_o.detach()), L.f32(rays_d.detach()), L.f32(z_vals)
        R, S = z.shape
        pts = torch.empty((R, S, 3), dtype=torch.float32, device=z.device)
        L.check(L.lib().us_ray_points(L.ptr(o), L.ptr(d), L.ptr(z), bhost, R, S, L.ptr(pts), L.stream()), "us_ray_points")
        ctx.bhost = bhost
        ctx.save_for_backward(z)
        return pts

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        R, S = z.shape
        g = L.f32(g)
        go = torch.empty((R, 3), dtype=torch.float32, device=z.device) if ctx.needs_input_grad[0] else None
        gd = torch.empty((R, 3), dtype=torch.float32, device=z.device) if ctx.needs_input_grad[1] else None
        if go is not None or gd is not None:
            L.check(L.lib().us_ray_points_bwd(L.ptr(g), L.ptr(z), ctx.bhost, R, S, L.ptr(go), L.ptr(gd), L.stream()),
                    "us_ray_points_bwd")
        return go, gd, None, None


class _SamplePointsFn(torch.autograd.Function):
    """
    Renderer.py:81-101 + :132-137 for a batch whose rays all carry a depth, ONE launch (us_sample_points): sorted + jittered z_vals
    [R,S] (no gradient) and the unit-cube points [R,S,3] (gradient to rays_o / rays_d: us_ray_points_bwd).  t_rand [R,S] replaces the
    jitter's draw; None: the in-kernel counter-based generator with `seed` (the reference draws torch.rand there, Renderer.py:54).
    """

    @staticmethod
    def forward(ctx, rays_o, rays_d, gt_depth, bhost, t_uni, t_surf, truncation, t_rand, seed, perturb):
        o, d, gd = L.f32(rays_o.detach()), L.f32(rays_d.detach()), L.f32(gt_depth.detach()).reshape(-1)
        R, S = o.shape[0], t_uni.shape[0] + t_surf.shape[0]
        dev = o.device
        z = torch.empty((R, S), dtype=torch.float32, device=dev)
        pts = torch.empty((R, S, 3), dtype=torch.float32, device=dev)
        valid = torch.empty(R, dtype=torch.uint8, device=dev)
        tr = L.f32(t_rand) if (perturb and t_rand is not None) else None
        L.check(L.lib().us_sample_points(L.ptr(o), L.ptr(d), L.ptr(gd), bhost, R, L.ptr(t_uni), t_uni.shape[0], L.ptr(t_surf),
                                         t_surf.shape[0], ctypes.c_float(1.2), ctypes.c_float(1.5 * truncation),
                                         ctypes.c_float(3 * truncation), L.ptr(tr), int(seed) & (2 ** 64 - 1), None, 1 if perturb else 0, 0,
                                         L.ptr(valid), L.ptr(z), L.ptr(pts), L.stream()), "us_sample_points")
        ctx.bhost = bhost
        ctx.save_for_backward(z)
        ctx.mark_non_differentiable(z)
        return z, pts

    @staticmethod
    def backward(ctx, _gz, g):
        (z,) = ctx.saved_tensors
        R, S = z.shape
        go = gd = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            g = L.f32(g)
            go = torch.empty((R, 3), dtype=torch.float32, device=z.device)
            gd = torch.empty((R, 3), dtype=torch.float32, device=z.device)
            L.check(L.lib().us_ray_points_bwd(L.ptr(g), L.ptr(z), ctx.bhost, R, S, L.ptr(go), L.ptr(gd), L.stream()),
                    "us_ray_points_bwd")
        return (go if ctx.needs_input_grad[0] else None, gd if ctx.needs_input_grad[1] else None) + (None,) * 8


class _CompositeFn(torch.autograd.Function):
    """raw[R,S,4], z[R,S], beta[1] -> term, pixel_unc, depth, rgb, depth_unc, sdf   (Renderer.py:140-152).  sdf [R,S] is the 4th channel
    of raw as render_batch_ray returns it: an output of THIS node (a strided view of raw's storage), so that the gradient the loss sends
    to it is added to d_raw inside us_composite_bwd instead of through a slice-backward (zeros + copy + add)."""

    @staticmethod
    def forward(ctx, raw, z_vals, beta):
        raw, z, b = L.f32(raw.detach()), L.f32(z_vals), L.f32(beta.detach()).reshape(1)
        R, S = z.shape
        dev = z.device
        term = torch.empty(R, device=dev); unc = torch.empty(R, device=dev); depth = torch.empty(R, device=dev)
        rgb = torch.empty((R, 3), device=dev); dunc = torch.empty(R, device=dev)
        L.check(L.lib().us_composite_fwd(L.ptr(raw), L.ptr(z), L.ptr(b), R, S, L.ptr(term), L.ptr(unc), L.ptr(depth),
                                         L.ptr(rgb), L.ptr(dunc), None, L.stream()), "us_composite_fwd")
        ctx.save_for_backward(raw, z, b)
        return term, unc, depth, rgb, dunc, raw.view(R, S, 4)[..., 3]

    @staticmethod
    def backward(ctx, g_term, g_unc, g_depth, g_rgb, g_dunc, g_sdf):
        raw, z, b = ctx.saved_tensors
        R, S = z.shape
        c = lambda t: None if t is None else L.f32(t)
        g_term, g_unc, g_depth, g_rgb, g_dunc, g_sdf = c(g_term), c(g_unc), c(g_depth), c(g_rgb), c(g_dunc), c(g_sdf)
        d_raw = torch.empty_like(raw)
        d_beta = torch.zeros(1, device=z.device) if ctx.needs_input_grad[2] else None
        part = torch.empty(R, device=z.device) if d_beta is not None else None
        L.check(L.lib().us_composite_bwd(L.ptr(raw), L.ptr(z), L.ptr(b), R, S, L.ptr(g_term), L.ptr(g_unc),
                                         L.ptr(g_depth), L.ptr(g_rgb), L.ptr(g_dunc), L.ptr(g_sdf), L.ptr(d_raw),
                                         L.ptr(d_beta), L.ptr(part), L.stream()), "us_composite_bwd")
        return d_raw, None, d_beta


def sample_z(gt_depth, truncation, t_uni, t_surf, t_rand=None):
    """Renderer.py:86-101 for rays with depth > 0: z_vals [R, n_strat+n_imp] (no gradient)."""
    gt = L.f32(gt_depth.detach()).reshape(-1)
    R = gt.shape[0]
    S = t_uni.shape[0] + t_surf.shape[0]
    z = torch.empty((R, S), dtype=torch.float32, device=gt.device)
    if R == 0:
        return z
    tr = None if t_rand is None else L.f32(t_rand)
    L.check(L.lib().us_sample_z(L.ptr(gt), R, L.ptr(t_uni), t_uni.shape[0], L.ptr(t_surf), t_surf.shape[0],
                                ctypes.c_float(1.2), ctypes.c_float(1.5 * truncation), ctypes.c_float(3 * truncation),
                                L.ptr(tr), L.ptr(z), L.stream()), "us_sample_z")
    return z


def _perturb(z_vals):
    """Renderer.py:42-57 in torch ops (the depth-guided path applies the jitter inside us_sample_z)."""
    mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    t_rand = torch.rand(z_vals.shape, device=z_vals.device)
    return lower + (upper - lower) * t_rand


def sdf2alpha(sdf, beta=10):
    """Renderer.py:154-158"""
    return 1. - torch.exp(-beta * torch.sigmoid(-sdf * beta))


def zero_depth_z(scene_rep, decoders, rays_o_uni, rays_d_uni, bound, t_uni, n_importance, perturb, device):
    """
    Renderer.py:104-130: rays WITHOUT a depth measurement get a coarse uniform pass (SDF grid + SDF decoder on the
    HIP kernels), then inverse-CDF samples from the resulting weights (common.sample_pdf, pdf left un-normalised
    as in the reference) merged into the uniform ones.  No gradient.  Returns z [R0, n_strat + n_imp].
    """
    with torch.no_grad():
        # far_bb + 0.01, z = far * t_uni, jitter (the draw of Renderer.py:54), normalize_3d_coordinate: one launch (us_uniform_points)
        o, d = L.f32(rays_o_uni.detach()), L.f32(rays_d_uni.detach())
        R0, Su = o.shape[0], t_uni.numel()
        tr = torch.rand((R0, Su), device=device) if perturb else None
        z_vals_uni = torch.empty((R0, Su), dtype=torch.float32, device=device)
        pts_uni_nor = torch.empty((R0 * Su, 3), dtype=torch.float32, device=device)
        L.check(L.lib().us_uniform_points(L.ptr(o), L.ptr(d), None, R0, bound_host(bound), L.ptr(L.f32(t_uni)), Su,
                                          L.ptr(L.f32(tr)) if tr is not None else None, 0, 1 if perturb else 0, L.ptr(z_vals_uni),
                                          L.ptr(pts_uni_nor), L.stream()), "us_uniform_points")
        sdf_uni = decoders.get_raw_sdf(pts_uni_nor, scene_rep).reshape(R0, Su)
        # alpha -> weights -> un-normalised cdf -> inverse transform -> merge sort: one HIP launch (us_importance_z)
        u = torch.rand([R0, n_importance], device=device)                  # the draw of common.sample_pdf (:61)
        beta = decoders.beta
        beta_t = L.f32(beta.detach()).reshape(1) if torch.is_tensor(beta) else torch.tensor([float(beta)], device=device)
        out = torch.empty((R0, Su + n_importance), dtype=torch.float32, device=device)
        L.check(L.lib().us_importance_z(L.ptr(L.f32(sdf_uni)), L.ptr(L.f32(z_vals_uni)), L.ptr(beta_t), L.ptr(u.contiguous()), R0, Su,
                                        n_importance, L.ptr(out), L.stream()), "us_importance_z")
        z_vals_uni = out
    return z_vals_uni


class Renderer(object):
    """
    Args (same as the reference): cfg (dict), unislam (object with bound, device, H, W, fx, fy, cx, cy),
    ray_batch_size (int).
    """

    def __init__(self, cfg, unislam, ray_batch_size=10000):
        self.ray_batch_size = ray_batch_size
        self.cfg = cfg
        self.perturb = cfg['rendering']['perturb']
        self.n_stratified = cfg['rendering']['n_stratified']
        self.n_importance = cfg['rendering']['n_importance']
        self.scale = cfg['scale']
        self.device = unislam.device
        self.bound = unislam.bound.to(unislam.device, non_blocking=True)
        self._bhost = bound_host(unislam.bound)
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = unislam.H, unislam.W, unislam.fx, unislam.fy, unislam.cx, unislam.cy
        # torch.linspace evaluated once on the CPU (the values the reference's CPU/CUDA linspace produce) and kept
        # on the device: Renderer.py:83-84 rebuilds them every call
        self._t_uni = torch.linspace(0., 1., steps=self.n_stratified).to(self.device)
        self._t_surf = torch.linspace(0., 1., steps=self.n_importance).to(self.device)
        # the jitter of a batch whose rays all carry a depth is drawn inside the sampling launch (a counter-based generator) whose seed is
        # drawn PER CALL from torch's default CPU generator (r6): torch.manual_seed() governs it whenever it is called, and two Renderers
        # (the tracker's and the mapper's copies of one pickled renderer) do not repeat each other's jitter

    def perturbation(self, z_vals):
        """Renderer.py:42-57"""
        return _perturb(z_vals)

    def sdf2alpha(self, sdf, beta=10):
        """Renderer.py:154-158"""
        return sdf2alpha(sdf, beta)

    def _zero_depth_z(self, scene_rep, decoders, rays_o_uni, rays_d_uni, device):
        return zero_depth_z(scene_rep, decoders, rays_o_uni, rays_d_uni, self._bhost, self._t_uni, self.n_importance,
                            self.perturb, device)

    def render_batch_ray(self, scene_rep, decoders, rays_d, rays_o, device, truncation, gt_depth=None, t_rand=None, all_depth=None):
        """
        Renderer.py:59-152.  Returns (termination_prob, pixel_unc, rendered_depth, rendered_rgb, sdf[R,S],
        z_vals[R,S], rendered_depth_uncertainty).  `t_rand` ([R_with_depth, S], optional) replaces the
        torch.rand draw of the jitter (parity tests); otherwise it is drawn on the device.  `all_depth` (optional): the caller already
        knows whether every ray carries a depth (render_img asks once per image instead of once per chunk).
        """
        n_rays = rays_o.shape[0]
        S = self.n_stratified + self.n_importance
        gt_depth = gt_depth.reshape(-1, 1)
        # the reference synchronises here too (Renderer.py:104: `if not gt_mask.all()`); one reduction instead of compare + all
        if all_depth is None:
            all_depth = n_rays == 0 or float(gt_depth.detach().min()) > 0
        gt_mask = None if all_depth else (gt_depth > 0).squeeze(-1)
        if all_depth and n_rays > 0:
            # z sampling + jitter + points in ONE launch (us_sample_points)
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()) if (self.perturb and t_rand is None) else 0   # (a CPU draw: no device sync)
            z_vals, pts = _SamplePointsFn.apply(rays_o, rays_d, gt_depth, self._bhost, self._t_uni, self._t_surf, float(truncation),
                                                t_rand, seed, bool(self.perturb))
            return self._decode_composite(scene_rep, decoders, pts, z_vals, device)
        if all_depth:
            z_vals = sample_z(gt_depth, truncation, self._t_uni, self._t_surf, None)
        else:
            z_vals = torch.empty([n_rays, S], device=device)
            gt_nonzero = gt_depth[gt_mask]
            if self.perturb and t_rand is None:
                t_rand = torch.rand((gt_nonzero.shape[0], S), device=device)
            z_vals[gt_mask] = sample_z(gt_nonzero, truncation, self._t_uni, self._t_surf, t_rand if self.perturb else None)
            z_vals[~gt_mask] = self._zero_depth_z(scene_rep, decoders, rays_o[~gt_mask].detach(),
                                                  rays_d[~gt_mask].detach(), device)
        pts = _RayPointsFn.apply(rays_o, rays_d, z_vals, self._bhost)          # normalised to [0,1] (Renderer.py:137)
        return self._decode_composite(scene_rep, decoders, pts, z_vals, device)

    def _decode_composite(self, scene_rep, decoders, pts, z_vals, device):
        """Renderer.py:139-152"""
        raw = decoders(pts, scene_rep)
        beta = decoders.beta if torch.is_tensor(decoders.beta) else torch.tensor([float(decoders.beta)], device=device)
        term, unc, depth, rgb, dunc, sdf = _CompositeFn.apply(raw, z_vals, beta)
        return term, unc, depth, rgb, sdf, z_vals, dunc

    def render_img(self, scene_rep, decoders, c2w, truncation, device, gt_depth=None):
        """Renderer.py:160-223: chunked forward-only render of a whole image."""
        with torch.no_grad():
            H, W = self.H, self.W
            rays_o, rays_d = get_rays(H, W, self.fx, self.fy, self.cx, self.cy, c2w, device)
            rays_o = rays_o.reshape(-1, 3); rays_d = rays_d.reshape(-1, 3)
            outs = [[], [], [], [], []]
            gt_depth = gt_depth.reshape(-1)
            # one question per image instead of one host synchronisation per chunk (Renderer.py:104 asks in every render_batch_ray call):
            # if every pixel has a depth, every chunk does; otherwise each chunk finds out for itself
            whole = True if (gt_depth.numel() > 0 and float(gt_depth.min()) > 0) else None
            for i in range(0, rays_d.shape[0], self.ray_batch_size):
                ret = self.render_batch_ray(scene_rep, decoders, rays_d[i:i + self.ray_batch_size].contiguous(),
                                            rays_o[i:i + self.ray_batch_size].contiguous(), device, truncation,
                                            gt_depth=gt_depth[i:i + self.ray_batch_size], all_depth=whole)
                term, unc, depth, color, _, _, dunc = ret
                outs[0].append(term.double()); outs[1].append(unc.double()); outs[2].append(dunc.double())
                outs[3].append(depth.double()); outs[4].append(color)
            term, unc, dunc, depth, color = [torch.cat(o, dim=0) for o in outs]
            return (depth.reshape(H, W), color.reshape(H, W, 3), term.reshape(H, W), unc.reshape(H, W),
                    dunc.reshape(H, W))
