"""
Forward-only dense SDF / colour query with the call contract of reference Mesher.eval_points
(src/utils/Mesher.py:134-166): the consumer of encoder + decoders that marching cubes is run on.  Only this query is
provided (the marching-cubes / open3d / trimesh part of the Mesher is host-side tooling outside the hot path).
"""
import torch


def eval_points(p, scene_rep, decoders, bound, points_batch_size=500000):
    """
    p [N,3] world coordinates -> [N,4] (rgb, sdf); points outside `bound` get sdf = -1 (Mesher.py:151-161).
    Runs under no_grad in chunks of points_batch_size (Mesher.py:145).
    """
    bound = bound.to(p)
    rets = []
    with torch.no_grad():
        for pi in torch.split(p, points_batch_size):
            mask = ((pi < bound[:, 1]) & (pi > bound[:, 0])).all(dim=-1)
            pn = (pi - bound[:, 0]) / (bound[:, 1] - bound[:, 0])
            ret = decoders(pn, scene_rep)
            ret[~mask, 3] = -1
            rets.append(ret)
    return torch.cat(rets, dim=0)
