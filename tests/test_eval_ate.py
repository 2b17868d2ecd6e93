"""unislam_amd.eval_ate against the fixture produced by the reference's src/tools/eval_ate.py (g12).  CPU only."""
import numpy as np
import pytest
import torch

from unislam_amd import eval_ate as E

KEYS = ("compared_pose_pairs", "error.rmse", "error.mean", "error.median", "error.std", "error.max")


@pytest.mark.parametrize("name", ["noisy", "drift", "planar"])
def test_g12_align_and_evaluate(golden, name):
    g = golden("g12_ate")
    gt, est = g[f"{name}_gt"], g[f"{name}_est"]
    rot, trans, err = E.align(est, gt)
    np.testing.assert_allclose(rot, g[f"{name}_rot"], atol=1e-9)
    np.testing.assert_allclose(trans, g[f"{name}_trans"], atol=1e-9)
    np.testing.assert_allclose(err, g[f"{name}_err"], atol=1e-9)
    assert abs(np.linalg.det(rot) - 1) < 1e-9
    n = gt.shape[1]
    first, second = {i: gt[:, i] for i in range(n)}, {i: est[:, i] for i in range(n)}
    for pa in (0, 1):
        te, res = E.evaluate_ate(first, second, pose_alignment=bool(pa))
        np.testing.assert_allclose(te, g[f"{name}_te{pa}"], atol=1e-7)
        assert [res[k] for k in KEYS] == list(g[f"{name}_res{pa}"]) and res["unit"] == "cm"
    # the error is invariant to a rigid motion of the estimate, and vanishes for a rigidly moved copy
    c, s = np.cos(0.7), np.sin(0.7)
    Rm = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    _, _, err2 = E.align(Rm @ est + np.array([[1.0], [2.0], [3.0]]), gt)
    np.testing.assert_allclose(err2, err, atol=1e-9)
    assert E.align(Rm @ gt + 0.5, gt)[2].max() < 1e-9


def test_g12_associate_and_pose_evaluation(golden):
    g = golden("g12_ate")
    a = {float(k): i for i, k in enumerate(g["assoc_a"])}
    b = {float(k): i for i, k in enumerate(g["assoc_b"])}
    np.testing.assert_array_equal(np.array(E.associate(a, b, 0.0, 0.02)), g["assoc_m"])
    np.testing.assert_array_equal(np.array(E.associate(a, b, -0.01, 0.02)), g["assoc_m_off"])
    gt, est = torch.from_numpy(g["pe_gt"]), torch.from_numpy(g["pe_est"])
    gt0 = gt.clone()
    te, res = E.pose_evaluation(gt, est, scale=2.0)
    np.testing.assert_allclose(te, g["pe_te"], atol=1e-5)
    assert [res[k] for k in KEYS] == list(g["pe_res"]) and res["compared_pose_pairs"] == 10
    assert torch.equal(torch.nan_to_num(gt), torch.nan_to_num(gt0))                   # inputs are left alone
    with pytest.raises(ValueError):
        E.evaluate_ate({0: [0, 0, 0]}, {5: [0, 0, 0]})
