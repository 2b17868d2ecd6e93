"""
Losses of the mapping / tracking iteration on the HIP kernels (csrc/render.hip K6):

  sdf_losses(...)      reference Mapper.sdf_losses / Tracker.sdf_losses (src/Mapper.py:141-175 == src/Tracker.py:113-147)
  mapping_loss(...)    the loss expression of src/Mapper.py:411-440 (both m_mask_mode values)
  tracking_loss(...)   the loss expression of src/Tracker.py:206-238 (both t_mask_mode values)

The reference selects rays with boolean indexing (dynamic shapes, one host sync per mask) and takes torch.mean of
each selection.  Here every term is kept as (sum, count): us_loss_stats produces the 5 sums and 5 counts with a
fixed-order reduction, us_loss_grad turns them into gradients.  An empty selection gives 0/0 = NaN exactly like
torch.mean of an empty tensor.  With `group` set, sums and counts are all-reduced over the ranks between the two
phases, so N ranks each holding a slice of the rays produce the gradient of the single-process loss.
"""
import ctypes

import torch

from . import _lib as L

MAP_ORIGINAL, MAP_NOMASK, TRK_ORIGINAL, TRK_NOMASK = 0, 1, 2, 3
_MODES = {("mapping", "original"): MAP_ORIGINAL, ("mapping", "no_mask"): MAP_NOMASK,
          ("tracking", "original"): TRK_ORIGINAL, ("tracking", "no_mask"): TRK_NOMASK}


def _sdf_arg(sdf):
    """sdf [R,S]: either contiguous or the 4th channel view of raw[R,S,4] (stride 4) -> (tensor, ptr, stride)"""
    S = sdf.shape[1]
    if sdf.stride(1) == 4 and sdf.stride(0) == 4 * S and sdf.dtype == torch.float32 and sdf.is_cuda:
        return sdf, ctypes.c_void_p(sdf.data_ptr()), 4
    t = L.f32(sdf)
    return t, L.ptr(t), 1


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, depth, rgb, z_vals, gt_depth, gt_color, pixel_unc, median, valid, mode, truncation, w5, group):
        sdf_t, sdf_p, sdf_stride = _sdf_arg(sdf.detach())
        depth_, rgb_ = L.f32(depth.detach()), L.f32(rgb.detach())
        z, gd, gc, unc = L.f32(z_vals), L.f32(gt_depth), L.f32(gt_color), L.f32(pixel_unc.detach())
        R, S = z.shape
        dev = z.device
        partials = torch.empty(int(L.lib().us_loss_partials_size(R)), dtype=torch.float32, device=dev)
        stats = torch.empty(10, dtype=torch.float32, device=dev)
        med = None if median is None else L.f32(median.detach()).reshape(1)
        # (a bool mask is one byte per ray: reinterpreted, not converted)
        val = None if valid is None else (valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8).contiguous())
        L.check(L.lib().us_loss_stats(mode, sdf_p, sdf_stride, L.ptr(val), L.ptr(z), L.ptr(gd), L.ptr(gc), L.ptr(depth_),
                                      L.ptr(rgb_), L.ptr(unc), L.ptr(med), R, S, float(truncation), L.ptr(partials),
                                      L.ptr(stats), L.stream()), "us_loss_stats")
        if group is not None:
            torch.distributed.all_reduce(stats, group=group if group is not True else None)
        # the three gradients in ONE buffer: the backward pass scales them by the upstream gradient with one multiplication
        g_all = torch.empty(R * S + 4 * R, dtype=torch.float32, device=dev)
        g_sdf, g_depth, g_rgb = g_all[:R * S].view(R, S), g_all[R * S:R * S + R], g_all[R * S + R:].view(R, 3)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        L.check(L.lib().us_loss_grad(mode, sdf_p, sdf_stride, L.ptr(val), L.ptr(z), L.ptr(gd), L.ptr(gc), L.ptr(depth_),
                                     L.ptr(rgb_), L.ptr(unc), L.ptr(med), R, S, float(truncation), L.host_floats(w5),
                                     L.ptr(stats), L.ptr(g_sdf), L.ptr(g_depth), L.ptr(g_rgb), L.ptr(loss), L.stream()),
                "us_loss_grad")
        ctx.save_for_backward(g_all)
        ctx.shape = (R, S)
        ctx.mark_non_differentiable(stats)
        return loss.reshape(()), stats

    @staticmethod
    def backward(ctx, g, _gs):
        (g_all,) = ctx.saved_tensors
        R, S = ctx.shape
        ga = g_all * g
        return (ga[:R * S].view(R, S), ga[R * S:R * S + R], ga[R * S + R:].view(R, 3)) + (None,) * 10


def fused_loss(kind, mask_mode, sdf, z_vals, depth, rgb, pixel_unc, gt_depth, gt_color, truncation, w, valid=None,
               group=None, return_stats=False):
    """
    loss = w_fs*fs + w_center*center + w_tail*tail + w_color*colour + w_depth*depth with the reference's ray gates.
    kind: 'mapping' | 'tracking';  mask_mode: 'original' | 'no_mask';  w: dict(fs, center, tail, color, depth).
    """
    mode = _MODES[(kind, mask_mode)]
    median = None
    if mode == TRK_ORIGINAL:
        median = (gt_depth - depth.detach()).abs().median()           # Tracker.py:214-215
    w5 = [w["fs"], w["center"], w["tail"], w["color"], w["depth"]]
    loss, stats = _LossFn.apply(sdf, depth, rgb, z_vals, gt_depth, gt_color, pixel_unc, median, valid, mode,
                                truncation, w5, group)
    return (loss, stats) if return_stats else loss


def mapping_loss(ret, gt_depth, gt_color, truncation, w, mask_mode="original", valid=None, group=None):
    """src/Mapper.py:411-440 on the 7-tuple `ret` of Renderer.render_batch_ray."""
    _, pixel_unc, depth, color, sdf, z_vals, _ = ret
    return fused_loss("mapping", mask_mode, sdf, z_vals, depth, color, pixel_unc, gt_depth, gt_color, truncation, w,
                      valid, group)


def tracking_loss(ret, gt_depth, gt_color, truncation, w, mask_mode="original", valid=None):
    """src/Tracker.py:206-238 on the 7-tuple `ret` of Renderer.render_batch_ray."""
    _, pixel_unc, depth, color, sdf, z_vals, _ = ret
    return fused_loss("tracking", mask_mode, sdf, z_vals, depth, color, pixel_unc, gt_depth, gt_color, truncation, w, valid)


def sdf_losses(sdf, z_vals, gt_depth, truncation, w_sdf_fs, w_sdf_center, w_sdf_tail):
    """
    src/Mapper.py:141-175: free-space / centre / tail SDF losses of the rays handed in (the caller has already
    selected them, as the reference does with boolean indexing).
    """
    R = z_vals.shape[0]
    dev = z_vals.device
    zeros1 = torch.zeros(R, device=dev)
    zeros3 = torch.zeros((R, 3), device=dev)
    w = dict(fs=w_sdf_fs, center=w_sdf_center, tail=w_sdf_tail, color=0.0, depth=0.0)
    return fused_loss("mapping", "no_mask", sdf, z_vals, zeros1, zeros3, zeros1, gt_depth, zeros3, truncation, w)
