"""
CPU, world_size 2, gloo: the data-parallel orchestration of unislam_amd.dist.dp_iterate (local sums/counts ->
all-reduce -> backward scaled by GLOBAL counts -> ONE gradient all-reduce(sum) -> Adam) gives the parameters of a
single process run on the concatenated batch.  The compute engine here is the CPU oracle (test infrastructure)
behind the same engine protocol MapStep implements on the HIP kernels.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import unislam_oracle as O

BOUND = O.load_bound([[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]])
W = dict(fs=5, center=200, tail=10, color=5, depth=0.1)
WK = ["fs", "center", "tail", "color", "depth"]
ECFG = {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 9, "base_resolution": 16,
        "per_level_scale": 1.4}


class OracleEngine:
    """engine protocol of unislam_amd.dist on the CPU oracle"""

    def __init__(self):
        torch.manual_seed(0)
        self.dec = O.DecodersOracle(c_dim=16)
        self.es, self.ec = O.HashGridOracle(3, ECFG), O.HashGridOracle(3, ECFG)
        with torch.no_grad():
            self.es.params.mul_(3000); self.ec.params.mul_(3000)
        self.params = list(self.dec.parameters()) + [self.es.params, self.ec.params]
        self.opt = torch.optim.Adam([{"params": list(self.dec.parameters()), "lr": 0.001},
                                     {"params": [self.es.params], "lr": 0.05}, {"params": [self.ec.params], "lr": 0.05}])
        self.stats = torch.zeros(10)
        self.grad = torch.zeros(sum(p.numel() for p in self.params))

    def forward(self, ro, rd, gd, gc, t_rand):
        ret = O.render_batch_ray(([self.es], [self.ec]), self.dec, rd, ro, 0.06, gd, BOUND, 32, 8, True, {"z": t_rand})
        _, unc, depth, color, sdf, z, _ = ret
        m = (gd > 0) & ((1 - unc.detach()) > 0.99)                       # Mapper.py:414-419
        front, _, center, tail = O.sdf_loss_masks(z[m], gd[m], 0.06)
        pred = z[m] + sdf[m] * 0.06
        g = gd[m][:, None].expand(z[m].shape)
        sums = [((sdf[m][front] - 1.0) ** 2).sum(), ((pred[center] - g[center]) ** 2).sum(),
                ((pred[tail] - g[tail]) ** 2).sum(), ((gc - color) ** 2).sum(), ((gd[m] - depth[m]) ** 2).sum()]
        counts = [front.sum(), center.sum(), tail.sum(), torch.tensor(gc.numel()), m.sum()]
        self._sums = sums
        self.stats = torch.stack([s.detach() for s in sums] + [c.float() for c in counts])

    def backward(self, on_ready=None):
        loss_local = sum(W[k] * self._sums[i] / self.stats[5 + i] for i, k in enumerate(WK))   # GLOBAL counts
        self.opt.zero_grad()
        loss_local.backward()
        self.grad = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])
        if on_ready is not None:                        # announce two segments, like MapStep does (colour table first)
            h = self.grad.numel() // 2
            on_ready(self.grad[h:]); on_ready(self.grad[:h])
        return sum(W[k] * self.stats[i] / self.stats[5 + i] for i, k in enumerate(WK))

    def adam_step(self):
        o = 0
        for p in self.params:
            p.grad = self.grad[o:o + p.numel()].view_as(p).clone(); o += p.numel()
        self.opt.step()

    def flat(self):
        return torch.cat([p.detach().reshape(-1) for p in self.params])


class ShardedOracleEngine(OracleEngine):
    """the same engine with flat parameter / moment buffers and adam_step(ranges): what dist._finish_sharded drives"""
    sharded_adam = True

    def __init__(self, world=2):
        super().__init__()
        n = sum(p.numel() for p in self.params)
        self.n = n
        q = 2 * world
        self.n_pad = (n + q - 1) // q * q                              # two announced halves, each divisible by the ranks
        self.flat = torch.zeros(self.n_pad); self.flat[:n] = super().flat()
        self.m, self.v, self.t = torch.zeros(self.n_pad), torch.zeros(self.n_pad), 0
        self.lr = torch.zeros(self.n_pad); o = 0
        for g in self.opt.param_groups:
            for p in g["params"]:
                self.lr[o:o + p.numel()] = g["lr"]; o += p.numel()

    def forward(self, *batch):
        o = 0
        with torch.no_grad():                                          # parameters <- the (all-gathered) flat buffer
            for p in self.params:
                p.copy_(self.flat[o:o + p.numel()].view_as(p)); o += p.numel()
        super().forward(*batch)

    def backward(self, on_ready=None):
        loss = super().backward(None)
        g = torch.zeros(self.n_pad); g[:self.n] = self.grad
        self.grad = g
        if on_ready is not None:
            h = self.n_pad // 2
            on_ready(self.grad[h:]); on_ready(self.grad[:h])
        return loss

    def adam_step(self, ranges=None):
        self.t += 1
        for (lo, hi) in (ranges or [(0, self.n_pad)]):
            g = self.grad[lo:hi]
            self.m[lo:hi] = 0.9 * self.m[lo:hi] + 0.1 * g
            self.v[lo:hi] = 0.999 * self.v[lo:hi] + 0.001 * g * g
            denom = (self.v[lo:hi].sqrt() / (1 - 0.999 ** self.t) ** 0.5) + 1e-8
            self.flat[lo:hi] -= self.lr[lo:hi] / (1 - 0.9 ** self.t) * self.m[lo:hi] / denom

    def flat_params(self):
        return self.flat[:self.n].clone()


def _worker_sharded(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from unislam_amd.dist import dp_iterate, init_from_env
    init_from_env(backend="gloo")
    eng = ShardedOracleEngine(world)
    losses = []
    for it in range(3):
        full = make_batch(48, 100 + it)
        losses.append(float(dp_iterate(eng, tuple(t[rank::world] for t in full), group=True)))
    flat = eng.flat_params()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        assert all(torch.equal(gathered[0], g) for g in gathered[1:])  # replicas stay bit-identical
        # the moments of the other ranks' shards were never touched
        h, q = eng.n_pad // 2, eng.n_pad // (2 * world)
        assert float(eng.m[h + q:].abs().max()) == 0.0 and float(eng.m[q:h].abs().max()) == 0.0
        np.savez(out_path, flat=flat.numpy(), losses=np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_sharded_adam_equal_one_process(world):
    """reduce-scatter + Adam on the rank's shard + all-gather (dist._finish_sharded; gloo: all-reduce stands in for the
    reduce-scatter) gives the parameters of one process with a dense torch.optim.Adam on the concatenated batch.
    world 8 = BASELINE config 4's rank count, rehearsed on the CPU."""
    from unislam_amd.dist import dp_iterate
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r0.npz")
        mp.spawn(_worker_sharded, args=(world, port, out), nprocs=world, join=True)
        res = np.load(out)
    eng = OracleEngine()
    losses = [float(dp_iterate(eng, make_batch(48, 100 + it), group=None)) for it in range(3)]
    np.testing.assert_allclose(res["losses"], losses, rtol=1e-5)
    np.testing.assert_allclose(res["flat"], eng.flat().numpy(), rtol=2e-4, atol=2e-6)


def make_batch(R, seed):
    g = torch.Generator().manual_seed(seed)
    ro = torch.tensor([[3.0, 1.2, 0.0]]).repeat(R, 1) + torch.randn(R, 3, generator=g) * 0.05
    rd = torch.randn(R, 3, generator=g); rd = rd / rd.norm(dim=-1, keepdim=True)
    gd = torch.rand(R, generator=g) * 2 + 0.4
    gd[::6] = 30.0                                   # uneven mask counts across the two halves
    gd[:R // 4] = 0.7
    return ro, rd, gd, torch.rand(R, 3, generator=g), torch.rand(R, 40, generator=g)


def _worker(rank, world, port, out_path, grad_comm=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from unislam_amd.dist import dp_iterate, init_from_env, shard_frames
    r, _, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and shard_frames(5, rank, world) == list(range(rank, 5, world))
    eng = OracleEngine()
    losses = []
    for it in range(2):
        full = make_batch(48, 100 + it)
        half = tuple(t[rank::world] for t in full)                    # ray slices: frames/rays shard naturally
        losses.append(float(dp_iterate(eng, half, group=True, grad_comm=grad_comm)))
    flat = eng.flat()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        assert all(torch.equal(gathered[0], g) for g in gathered[1:])  # replicas stay bit-identical
        np.savez(out_path, flat=flat.numpy(), losses=np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_equal_one_process(world):
    """world 8 = BASELINE config 4's rank count (8 frames per step, one per rank), rehearsed on the CPU"""
    from unislam_amd.dist import dp_iterate
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r0.npz")
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = np.load(out)
    eng = OracleEngine()
    losses = []
    for it in range(2):
        full = make_batch(48, 100 + it)
        # same ray order as the two slices concatenated does not matter: every term is a sum over rays
        losses.append(float(dp_iterate(eng, full, group=None)))
    np.testing.assert_allclose(res["losses"], losses, rtol=1e-5)
    np.testing.assert_allclose(res["flat"], eng.flat().numpy(), rtol=1e-4, atol=1e-6)


def test_two_ranks_bf16_gradient_payload():
    """grad_comm="bf16": the all-reduce carries bfloat16; replicas stay identical, parameters follow the fp32 run closely"""
    from unislam_amd.dist import dp_iterate
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r0.npz")
        mp.spawn(_worker, args=(2, port, out, "bf16"), nprocs=2, join=True)
        res = np.load(out)
    eng = OracleEngine()
    start = eng.flat().numpy().copy()
    for it in range(2):
        dp_iterate(eng, make_batch(48, 100 + it), group=None)
    ref = eng.flat().numpy()
    # Adam steps are bounded by lr per entry: compare the UPDATES (a sign flip of a tiny gradient may move one entry by 2 lr)
    du, dr = res["flat"] - start, ref - start
    assert np.linalg.norm(du - dr) < 0.1 * np.linalg.norm(dr)


def test_averaging_local_means_is_not_equivalent():
    """the naive scheme (each rank normalises by its own counts, gradients averaged) differs: documents why counts travel"""
    eng = OracleEngine()
    full = make_batch(48, 100)
    eng.forward(*full); g_full = None
    eng.backward(); g_full = eng.grad.clone()
    gs = []
    for r in range(2):
        e = OracleEngine()
        e.forward(*tuple(t[:12] if r == 0 else t[12:] for t in full)); e.backward(); gs.append(e.grad.clone())
    naive = 0.5 * (gs[0] + gs[1])
    assert (naive - g_full).abs().max() > 1e-3 * g_full.abs().max()     # uneven slices: 12 vs 36 rays


def _worker_budget_loop(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import time
    from unislam_amd.dist import init_from_env, all_agree
    init_from_env("gloo")
    # a loop with a collective in its body and a per-rank wall-clock budget (rank r may spend 0.05 * (r + 1) s): as bench.py's set-up phase
    t0, turns, acc = time.perf_counter(), 0, torch.zeros(1)
    while all_agree(time.perf_counter() - t0 < 0.05 * (rank + 1)):
        dist.all_reduce(acc)                                         # would hang if the ranks ran different numbers of turns
        acc += 1.0
        time.sleep(0.004)
        turns += 1
    flags = [all_agree(rank == 0), all_agree(True), all_agree(False)]
    torch.save({"turns": turns, "flags": flags}, f"{out_path}.{rank}")
    dist.destroy_process_group()


def test_all_agree_keeps_rank_local_budgets_in_step():
    """unislam_amd.dist.all_agree: a loop that every rank would leave by its own wall clock, with a collective in its body, runs the same
    number of turns on every rank (the shortest budget decides) -- the condition under which bench.py's set-up phase deadlocked a
    data-parallel run before it used this -- and a flag that is true on one rank only is false everywhere."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o")
        mp.spawn(_worker_budget_loop, args=(2, port, out), nprocs=2, join=True)
        r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["turns"] == r1["turns"] and 1 <= r0["turns"] <= 40
    assert r0["flags"] == r1["flags"] == [False, True, False]
    from unislam_amd.dist import all_agree
    assert all_agree(True, None) is True and all_agree(False, None) is False       # no process group: the flag itself


# ------------------------------------------------------------------------------------------------ joint_opt windows (src/Mapper.py:359-376)
class WindowOracleEngine(OracleEngine):
    """OracleEngine + the camera poses of THIS rank's share of a mapping window as one more (rank-local) Adam group: forward() builds the
    rays of its frames from the poses under autograd (src/Mapper.py:372-380), backward() leaves the pose gradients, pose_step() is what
    dist.dp_iterate runs as `before_adam`.  fixed_first: this rank holds the window's oldest frame, which stays fixed (:374)."""

    def __init__(self, c2ws, depths, colors, dirs, fixed_first):
        super().__init__()
        self.depths, self.colors, self.dirs = depths, colors, dirs
        self.c2w_first = c2ws[0] if fixed_first else None
        self.poses = torch.nn.Parameter(O.matrix_to_cam_pose(c2ws[(1 if fixed_first else 0):]))
        self.pose_opt = torch.optim.Adam([self.poses], lr=1e-3)

    def forward(self, idx, t_rand):
        mats = O.cam_pose_to_matrix(self.poses)
        c2ws = mats if self.c2w_first is None else torch.cat([self.c2w_first[None], mats], dim=0)
        ro, rd, gd, gc = O.get_samples_all(idx.shape[1], c2ws, self.depths, self.colors, self.dirs, idx)
        super().forward(ro, rd, gd, gc, t_rand)

    def backward(self, on_ready=None, ray_grads=False):
        self.pose_opt.zero_grad()
        return super().backward(on_ready)

    def pose_step(self):
        self.pose_opt.step()

    def c2ws(self):
        mats = O.cam_pose_to_matrix(self.poses.detach())
        return mats if self.c2w_first is None else torch.cat([self.c2w_first[None], mats], dim=0)


def make_window(B, P, seed):
    g = torch.Generator().manual_seed(seed)
    c2ws = []
    for _ in range(B):
        q = torch.randn(4, generator=g); q = q / q.norm()
        m = torch.eye(4); m[:3, :3] = O.quaternion_to_matrix(q[None])[0]; m[:3, 3] = torch.tensor([3.0, 1.2, 0.0]) + torch.randn(3, generator=g) * 0.05
        c2ws.append(m)
    dirs = torch.randn(B, P, 3, generator=g); dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    depths = torch.rand(B, P, generator=g) * 2 + 0.4
    depths[:, ::6] = 30.0
    return torch.stack(c2ws), depths, torch.rand(B, P, 3, generator=g), dirs


def _window_draws(B, P, n, it):
    g = torch.Generator().manual_seed(500 + it)
    return torch.randint(P, (B, n), generator=g), torch.rand(B, n, 40, generator=g)


def _worker_window(rank, world, port, out_path, B):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from unislam_amd.dist import dp_iterate, init_from_env, shard_frames
    init_from_env(backend="gloo")
    P, n = 40, 6
    c2ws, depths, colors, dirs = make_window(B, P, 77)
    own = shard_frames(B, rank, world)
    eng = WindowOracleEngine(c2ws[own], depths[own], colors[own], dirs[own], fixed_first=(own[0] == 0))
    losses = []
    for it in range(3):
        idx, tr = _window_draws(B, P, n, it)
        losses.append(float(dp_iterate(eng, (idx[own], tr[own].reshape(-1, 40)), group=True, ray_grads=True, before_adam=eng.pose_step)))
    flat = eng.flat()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    n_max = -(-B // world)
    mine = torch.zeros(n_max, 4, 4); mine[:len(own)] = eng.c2ws()
    poses = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(poses, mine)
    if rank == 0:
        assert all(torch.equal(gathered[0], g) for g in gathered[1:])  # the model replicas stay bit-identical
        full = torch.zeros(B, 4, 4)
        for k in range(world):
            fr = shard_frames(B, k, world)
            full[fr] = poses[k][:len(fr)]
        np.savez(out_path, flat=flat.numpy(), losses=np.array(losses), c2ws=full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 6), (8, 8)])
def test_ranks_equal_one_process_on_a_joint_opt_window(world, B):
    """the reference's default mapping iteration (joint_opt: the window's poses are an Adam group, src/Mapper.py:359-376,443-459) data-parallel:
    rank r owns the frames {f : f mod W == r} with their poses and pose moments; the step exchanges loss statistics and model gradients
    only (a frame's rays live on one rank, so its pose gradient is complete there).  Model parameters AND poses equal one process on the
    whole window; the oldest pose stays fixed."""
    from unislam_amd.dist import dp_iterate
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r0.npz")
        mp.spawn(_worker_window, args=(world, port, out, B), nprocs=world, join=True)
        res = np.load(out)
    P, n = 40, 6
    c2ws, depths, colors, dirs = make_window(B, P, 77)
    eng = WindowOracleEngine(c2ws, depths, colors, dirs, fixed_first=True)
    losses = []
    for it in range(3):
        idx, tr = _window_draws(B, P, n, it)
        losses.append(float(dp_iterate(eng, (idx, tr.reshape(-1, 40)), group=None, ray_grads=True, before_adam=eng.pose_step)))
    np.testing.assert_allclose(res["losses"], losses, rtol=1e-5)
    np.testing.assert_allclose(res["flat"], eng.flat().numpy(), rtol=1e-4, atol=1e-6)
    one = eng.c2ws().numpy()
    np.testing.assert_allclose(res["c2ws"], one, rtol=0, atol=2e-6)
    assert np.array_equal(res["c2ws"][0], c2ws[0].numpy())                        # "we fix the oldest c2w", src/Mapper.py:374
    assert np.abs(one[1:] - c2ws[1:].numpy()).max() > 1e-4                        # the others did move
