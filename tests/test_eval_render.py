"""
CPU: the render-quality report (unislam_amd.eval_render, after src/tools/eval_recon.py:235-307) on a stub renderer: the PSNR /
depth-L1 arithmetic, the gt_depth > 0 mask, the stride of 5 frames and the 4-decimal rounding of the reference's report.
"""
import math

import torch

from unislam_amd.eval_render import eval_rendering, psnr_and_depth_l1


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
