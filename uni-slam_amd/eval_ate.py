"""
Absolute trajectory error of an estimated camera trajectory (reference src/tools/eval_ate.py:169-236,270-281,380-506,508-552):
the number BASELINE config 5 is quoted in.  Horn's closed-form rigid alignment of the estimated onto the given positions, the
per-frame translational error after that alignment, and the summary the reference prints (in cm, rounded to 2 decimals).

numpy float64 throughout, plain arrays instead of numpy.matrix; no plotting.  `align` and `evaluate_ate` are pinned by
tests/golden/g12_ate.npz, produced by the reference's functions.
"""
import numpy as np
import torch


def associate(first, second, offset=0.0, max_difference=0.02):
    """
    Greedy closest-stamp matching of two {stamp: data} dictionaries (eval_ate.py:169-199): candidate pairs within
    max_difference, taken in order of increasing |a - (b + offset)|, each stamp used once; returned sorted by the first stamp.
    """
    cand = sorted((abs(a - (b + offset)), a, b) for a in first for b in second if abs(a - (b + offset)) < max_difference)
    free_a, free_b = set(first), set(second)
    matches = []
    for _, a, b in cand:
        if a in free_a and b in free_b:
            free_a.remove(a)
            free_b.remove(b)
            matches.append((a, b))
    matches.sort()
    return matches


def align(model, data):
    """
    Horn's method (eval_ate.py:202-236): rotation R [3,3] and translation t [3,1] minimising sum |R model_i + t - data_i|^2 over
    the columns of model, data [3,n]; returns (R, t, trans_error [n]) with trans_error_i = |R model_i + t - data_i|.
    """
    model, data = np.asarray(model, dtype=np.float64), np.asarray(data, dtype=np.float64)
    mm, dm = model.mean(1, keepdims=True), data.mean(1, keepdims=True)
    W = (model - mm) @ (data - dm).T                                   # sum of outer(model_i, data_i)
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:                       # keep a proper rotation
        S[2, 2] = -1
    rot = U @ S @ Vh
    trans = dm - rot @ mm
    err = rot @ model + trans - data
    return rot, trans, np.sqrt((err * err).sum(0))


def evaluate_ate(first, second, pose_alignment=False, offset=0.0, scale=1.0, max_difference=0.02):
    """
    first / second: {stamp: (x, y, z, ...)} given and estimated trajectory (eval_ate.py:380-506).  Returns (trans_error in cm [n],
    results dict with the reference's keys).  As in the reference, the error is ALWAYS the one after alignment -- the
    `pose_alignment` switch only chose which trajectory the reference plotted -- and is reported in cm.
    `aligned` in the result holds the estimated positions [n,3] (aligned if pose_alignment) for callers that want them.
    """
    matches = associate(first, second, float(offset), float(max_difference))
    if len(matches) < 2:
        raise ValueError("Couldn't find matching timestamp pairs between groundtruth and estimated trajectory!")
    first_xyz = np.array([[float(v) for v in first[a][0:3]] for a, b in matches], dtype=np.float64).T
    second_xyz = np.array([[float(v) * float(scale) for v in second[b][0:3]] for a, b in matches], dtype=np.float64).T
    rot, trans, trans_error = align(second_xyz, first_xyz)
    aligned = (rot @ second_xyz + trans) if pose_alignment else second_xyz
    trans_error = trans_error * 100
    n = len(trans_error)
    return trans_error, {
        "compared_pose_pairs": n,
        "unit": "cm",
        "error.rmse": round(float(np.sqrt(np.dot(trans_error, trans_error) / n)), 2),
        "error.mean": round(float(np.mean(trans_error)), 2),
        "error.median": round(float(np.median(trans_error)), 2),
        "error.std": round(float(np.std(trans_error)), 2),
        "error.max": round(float(np.max(trans_error)), 2),
        "aligned": aligned.T,
    }


def convert_poses(c2w_list, N, scale, gt=True):
    """
    c2w matrices [>= N+1, 4, 4] -> (poses [n,7] as (T, quaternion), mask [N+1]) (eval_ate.py:527-549): given poses holding inf / nan
    (ScanNet) are masked out; translations are divided by `scale` (on a copy; the reference divides its input in place).
    """
    from .common import matrix_to_cam_pose
    mask = torch.ones(N + 1).bool()
    poses = []
    for idx in range(N + 1):
        c2w = c2w_list[idx].detach().clone()
        if gt and (torch.isinf(c2w).any() or torch.isnan(c2w).any()):
            mask[idx] = 0
            continue
        c2w[:3, 3] /= scale
        poses.append(matrix_to_cam_pose(c2w.unsqueeze(0), RT=False))
    return torch.cat(poses, dim=0), mask


def pose_evaluation(gt_c2w_list, estimate_c2w_list, scale=1.0, pose_alignment=False):
    """eval_ate.py:270-281 + 508-525 without the plots: frame index as the time stamp, invalid given poses dropped"""
    N = len(gt_c2w_list) - 1
    poses_gt, mask = convert_poses(gt_c2w_list, N, scale)
    poses_est, _ = convert_poses(estimate_c2w_list, N, scale, gt=False)
    poses_est = poses_est[mask]
    g, e = poses_gt.cpu().numpy(), poses_est.cpu().numpy()
    return evaluate_ate({i: g[i] for i in range(g.shape[0])}, {i: e[i] for i in range(e.shape[0])}, pose_alignment=pose_alignment)
