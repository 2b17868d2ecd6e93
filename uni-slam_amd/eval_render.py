"""
Render-quality evaluation of a finished run (reference src/tools/eval_recon.py:235-307, `eval_rendering`): every `stride`-th frame
is re-rendered at its ESTIMATED pose with Renderer.render_img, and the averages of

    PSNR            -10 log10( mse(gt_color[gt_depth > 0], color[gt_depth > 0]) )        (:277-278)
    depth L1        mean |gt_depth - depth| over gt_depth > 0                             (:287)

    MS-SSIM         ms_ssim(gt_color, color, data_range=1.0) over the whole frame                 (:278-279)

are reported under the reference's keys `avg_psnr`, `depth_l1_render` and `avg_ms_ssim` (4 decimals, :294-302).  The reference takes
MS-SSIM from pytorch_msssim and LPIPS from torchmetrics; neither package is in this image.  `ms_ssim` below restates the published
algorithm (Wang, Simoncelli, Bovik 2003) with that package's conventions (11-tap Gaussian, sigma 1.5, no padding, five scales, 2 x 2
mean pooling, weights 0.0448 / 0.2856 / 0.3001 / 0.2363 / 0.1333, K = 0.01 / 0.03) -- PARITY UNPINNED: no vector of the package exists
here; tests/test_eval_render.py checks it against an independent scipy restatement.  Frames whose shorter side is <= 160 pixels have
no five-scale MS-SSIM (the package asserts there): the key is then absent.  LPIPS needs AlexNet weights: left out (key absent, not
zero).  Pure torch; the rendering itself runs on the HIP kernels through Renderer.render_img.
"""
import torch

MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_window(size=11, sigma=1.5, dtype=torch.float32, device="cpu"):
    x = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(x ** 2) / (2.0 * sigma ** 2))
    return g / g.sum()


def _blur(x, win):
    """separable, per channel, no padding: [B, C, H, W] -> [B, C, H - k + 1, W - k + 1]"""
    c, k = x.shape[1], win.numel()
    x = torch.nn.functional.conv2d(x, win.view(1, 1, k, 1).expand(c, 1, k, 1), groups=c)
    return torch.nn.functional.conv2d(x, win.view(1, 1, 1, k).expand(c, 1, 1, k), groups=c)


def _ssim_and_cs(x, y, win, data_range):
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _blur(x, win), _blur(y, win)
    s11 = _blur(x * x, win) - mu1 * mu1
    s22 = _blur(y * y, win) - mu2 * mu2
    s12 = _blur(x * y, win) - mu1 * mu2
    cs = (2.0 * s12 + c2) / (s11 + s22 + c2)
    ssim = (2.0 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1) * cs
    return ssim.flatten(2).mean(-1), cs.flatten(2).mean(-1)                     # [B, C] each


def ms_ssim(x, y, data_range=1.0, size_average=True):
    """multi-scale SSIM of two image batches [B, C, H, W] (the call of eval_recon.py:278-279)"""
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("ms_ssim: two [B, C, H, W] batches of one shape")
    if min(x.shape[2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("ms_ssim: the shorter image side must exceed 160 pixels for five scales")
    win = _gauss_window(dtype=x.dtype, device=x.device)
    terms = []
    for level in range(5):
        ssim, cs = _ssim_and_cs(x, y, win, data_range)
        if level < 4:
            terms.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = torch.nn.functional.avg_pool2d(x, kernel_size=2, padding=pad)
            y = torch.nn.functional.avg_pool2d(y, kernel_size=2, padding=pad)
    terms.append(torch.relu(ssim))
    w = torch.tensor(MS_SSIM_WEIGHTS, dtype=x.dtype, device=x.device).view(-1, 1, 1)
    val = torch.prod(torch.stack(terms, 0) ** w, dim=0)                          # [B, C]
    return val.mean() if size_average else val.mean(1)


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
    renderer: unislam_amd.Renderer.  Returns {"avg_psnr", "depth_l1_render", "frames"} and, for frames large enough, "avg_ms_ssim".
    """
    psnr_sum, l1_sum, ssim_sum, cnt, idx = 0.0, 0.0, 0.0, 0, 0
    while idx < n_img:                                                             # eval_recon.py:257 ... :289 (render_idx += 5)
        _, gt_color, gt_depth, _, _ = frame_reader[idx]
        gt_color = gt_color.squeeze(0).to(device, non_blocking=True)
        gt_depth = gt_depth.squeeze(0).to(device, non_blocking=True)
        with torch.no_grad():
            depth, color, _, _, _ = renderer.render_img(scene_rep, decoders, estimate_c2w_list[idx].to(device), truncation, device,
                                                        gt_depth=gt_depth)
        p, l = psnr_and_depth_l1(gt_color, gt_depth, color, depth)
        psnr_sum += p; l1_sum += l; cnt += 1
        if ssim_sum is not None and min(gt_color.shape[:2]) > 160:
            ssim_sum += float(ms_ssim(gt_color.transpose(0, 2).unsqueeze(0).float(), color.transpose(0, 2).unsqueeze(0).float(), data_range=1.0))
        else:
            ssim_sum = None
        idx += stride
    res = {"avg_psnr": float(f"{psnr_sum / cnt:.4f}"), "depth_l1_render": float(f"{l1_sum / cnt:.4f}"), "frames": cnt}
    if ssim_sum is not None:
        res["avg_ms_ssim"] = float(f"{ssim_sum / cnt:.4f}")
    return res
