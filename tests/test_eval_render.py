"""
CPU: the render-quality report (unislam_amd.eval_render, after src/tools/eval_recon.py:235-307) on a stub renderer: the PSNR /
depth-L1 arithmetic, the gt_depth > 0 mask, the stride of 5 frames and the 4-decimal rounding of the reference's report.
"""
import math

import torch

import numpy as np

from unislam_amd.eval_render import eval_rendering, ms_ssim, psnr_and_depth_l1


def test_psnr_and_depth_l1_formulas():
    g = torch.Generator().manual_seed(0)
    gt_c, gt_d = torch.rand(6, 8, 3, generator=g), torch.rand(6, 8, generator=g) + 0.5
    gt_d[0, :3] = 0.0                                                   # pixels without a depth measurement are left out
    c, d = gt_c + 0.1, gt_d + 0.02
    c[0, :3] += 5.0; d[0, :3] += 7.0                                    # ... whatever was rendered there
    psnr, l1 = psnr_and_depth_l1(gt_c, gt_d, c, d)
    assert abs(psnr - (-10 * math.log10(0.1 ** 2))) < 1e-4 and abs(l1 - 0.02) < 1e-6


def test_eval_rendering_stride_and_keys():
    class Frames:
        def __getitem__(self, i):
            return i, torch.full((1, 4, 4, 3), 0.5), torch.full((1, 4, 4), 1.0 + i), torch.eye(4), None

    seen = []

    class Rend:
        def render_img(self, scene_rep, decoders, c2w, truncation, device, gt_depth=None):
            seen.append(float(gt_depth[0, 0]))
            return gt_depth + 0.01 * len(seen), torch.full((4, 4, 3), 0.6), None, None, None

    res = eval_rendering(12, Frames(), torch.eye(4).repeat(12, 1, 1), Rend(), None, None, 0.06, "cpu")
    assert seen == [1.0, 6.0, 11.0] and res["frames"] == 3                # frames 0, 5, 10 (eval_recon.py:289)
    assert res["avg_psnr"] == 20.0 and abs(res["depth_l1_render"] - 0.02) < 1e-6


def _ms_ssim_scipy(x, y):
    """the published algorithm once more, on numpy / scipy.ndimage, one [C, H, W] pair"""
    from scipy.ndimage import correlate1d
    t = np.arange(11) - 5.0
    g = np.exp(-t ** 2 / (2 * 1.5 ** 2)); g /= g.sum()

    def blur(a):                                                        # 'valid' part of the separable filter
        a = correlate1d(correlate1d(a, g, axis=1, mode="constant"), g, axis=2, mode="constant")
        return a[:, 5:-5, 5:-5]

    def pool(a):
        c, h, w = a.shape
        a = np.pad(a, ((0, 0), (h % 2, h % 2), (w % 2, w % 2)))          # avg_pool2d(padding = s % 2), zeros counted
        h2, w2 = a.shape[1] // 2, a.shape[2] // 2
        return a[:, :2 * h2, :2 * w2].reshape(c, h2, 2, w2, 2).mean((2, 4))

    c1, c2, out = 0.01 ** 2, 0.03 ** 2, []
    for level in range(5):
        m1, m2 = blur(x), blur(y)
        s11, s22, s12 = blur(x * x) - m1 * m1, blur(y * y) - m2 * m2, blur(x * y) - m1 * m2
        cs = (2 * s12 + c2) / (s11 + s22 + c2)
        ss = (2 * m1 * m2 + c1) / (m1 * m1 + m2 * m2 + c1) * cs
        if level < 4:
            out.append(np.maximum(cs.mean((1, 2)), 0)); x, y = pool(x), pool(y)
    out.append(np.maximum(ss.mean((1, 2)), 0))
    w = np.array([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])[:, None]
    return float(np.prod(np.stack(out) ** w, 0).mean())


def test_ms_ssim_properties_and_an_independent_restatement():
    g = torch.Generator().manual_seed(3)
    yy, xx = torch.meshgrid(torch.linspace(0, 6, 177), torch.linspace(0, 9, 203), indexing="ij")
    a = torch.stack([0.5 + 0.4 * torch.sin(xx + k) * torch.cos(yy * (k + 1)) for k in range(3)])[None].double()
    b = (a + 0.05 * torch.randn(a.shape, generator=g, dtype=torch.float64)).clamp(0, 1)
    assert abs(float(ms_ssim(a, a)) - 1.0) < 1e-12                      # identical frames
    v = float(ms_ssim(a, b))
    assert 0.5 < v < 0.999 and abs(v - float(ms_ssim(b, a))) < 1e-12     # symmetric, below 1
    assert float(ms_ssim(a, (a + 0.15 * torch.randn(a.shape, generator=g, dtype=torch.float64)).clamp(0, 1))) < v   # more noise, lower
    assert abs(v - _ms_ssim_scipy(a[0].numpy(), b[0].numpy())) < 1e-10
    assert ms_ssim(torch.cat([a, b]), torch.cat([b, b]), size_average=False).shape == (2,)
    try:
        ms_ssim(a[..., :160, :], b[..., :160, :]); assert False
    except ValueError:
        pass


def test_eval_rendering_reports_ms_ssim_for_large_frames():
    class Frames:
        def __getitem__(self, i):
            return i, torch.full((1, 170, 180, 3), 0.5), torch.full((1, 170, 180), 1.0), torch.eye(4), None

    class Rend:
        def render_img(self, scene_rep, decoders, c2w, truncation, device, gt_depth=None):
            return gt_depth, torch.full((170, 180, 3), 0.5), None, None, None

    res = eval_rendering(6, Frames(), torch.eye(4).repeat(6, 1, 1), Rend(), None, None, 0.06, "cpu")
    assert res["avg_ms_ssim"] == 1.0 and res["frames"] == 2
