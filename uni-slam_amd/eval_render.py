"""
Render-quality evaluation of a finished run (reference src/tools/eval_recon.py:235-307, `eval_rendering`): every `stride`-th frame
is re-rendered at its ESTIMATED pose with Renderer.render_img, and the averages of

    PSNR            -10 log10( mse(gt_color[gt_depth > 0], color[gt_depth > 0]) )        (:277-278)
    depth L1        mean |gt_depth - depth| over gt_depth > 0                             (:287)

are reported under the reference's keys `avg_psnr` and `depth_l1_render` (4 decimals, :294-302).  The reference also reports
MS-SSIM and LPIPS from pytorch_msssim / torchmetrics; neither package is in this image and both are learned or library metrics
outside the hot path, so they are left out (their keys are absent, not zero).  Pure torch; the rendering itself runs on the HIP
kernels through Renderer.render_img.
"""
import torch


def psnr_and_depth_l1(gt_color, gt_depth, color, depth):
    """one frame: (psnr [dB], depth L1 [m]) over the pixels with a depth measurement (eval_recon.py:277-278,287)"""
    m = gt_depth > 0
    mse = torch.nn.functional.mse_loss(gt_color[m].float(), color[m].float())
    psnr = -10.0 * torch.log10(mse)
    l1 = torch.abs(gt_depth[m].float() - depth[m].float()).mean()
    return float(psnr), float(l1)


def eval_rendering(n_img, frame_reader, estimate_c2w_list, renderer, scene_rep, decoders, truncation, device, stride=5):
    """
    frame_reader[i] -> (idx, gt_color [H,W,3], gt_depth [H,W], gt_c2w, rays_d) as in the reference's datasets;
    renderer: unislam_amd.Renderer.  Returns {"avg_psnr", "depth_l1_render", "frames"}.
    """
    psnr_sum, l1_sum, cnt, idx = 0.0, 0.0, 0, 0
    while idx < n_img:                                                             # eval_recon.py:257 ... :289 (render_idx += 5)
        _, gt_color, gt_depth, _, _ = frame_reader[idx]
        gt_color = gt_color.squeeze(0).to(device, non_blocking=True)
        gt_depth = gt_depth.squeeze(0).to(device, non_blocking=True)
        with torch.no_grad():
            depth, color, _, _, _ = renderer.render_img(scene_rep, decoders, estimate_c2w_list[idx].to(device), truncation, device,
                                                        gt_depth=gt_depth)
        p, l = psnr_and_depth_l1(gt_color, gt_depth, color, depth)
        psnr_sum += p; l1_sum += l; cnt += 1
        idx += stride
    return {"avg_psnr": float(f"{psnr_sum / cnt:.4f}"), "depth_l1_render": float(f"{l1_sum / cnt:.4f}"), "frames": cnt}
