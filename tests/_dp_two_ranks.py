"""
Helper of tests/test_gpu_step.py::test_mapstep_ranks_on_one_gpu (not a test module): one data-parallel rank.
argv: rank world port out_path variant.  Both ranks use cuda:0 and the gloo backend (RCCL refuses two ranks on one device); the
MapStep / dist.dp_iterate code path is the one bench.py runs under torch.distributed.run.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
    sys.path.insert(0, p)


def main():
    rank, world, port, out, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import unislam_amd as us
    from unislam_amd.dist import broadcast_parameters
    import test_gpu_step as T
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if variant.startswith("window"):
            return window_rank(us, T, rank, world, out, variant)
        kw = {"plain": {}, "sharded": dict(sharded_adam=True), "bf16": dict(grad_comm="bf16")}[variant]
        dec, es, ec = T._scene(us, False, seed=11)
        R = 256
        step = us.MapStep(es, ec, dec, T.BOUND, 32, 8, 0.06, T.W, T.LR, max_rays=R, group=True, **kw)
        broadcast_parameters(step.flat)
        ro, rd, gd, gc = T._rays(R, seed=100 + rank, outside=(rank % 2 == 1))    # odd ranks also have rays the pre-filter drops
        t_rand = torch.rand(R, 40, generator=torch.Generator().manual_seed(200 + rank)).to(T.DEV)
        losses = [float(step.iterate(ro, rd, gd, gc, t_rand=t_rand, has_zero_depth=False)) for _ in range(3)]
        torch.cuda.synchronize()
        torch.save({"flat": step.flat.detach().cpu(), "losses": losses, "step_dev": float(step.step_dev[0])}, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


def window_draws(B, P, n_per, extra, S, it):
    g = torch.Generator().manual_seed(900 + it)
    idx = torch.randint(P, (B, n_per), generator=g)
    idx2 = torch.randint(P, (min(extra[0], B), extra[1]), generator=g) if extra else None
    tr = torch.rand(B, n_per, S, generator=g)
    tr2 = torch.rand(min(extra[0], B), extra[1], S, generator=g) if extra else None
    return idx, idx2, tr, tr2


def window_rank(us, T, rank, world, out, variant):
    """variants window / window_extra / window_graph / window_sharded: the reference's joint_opt iteration (src/Mapper.py:359-376,443-459)
    data-parallel -- MapWindow.sharded gives this rank the frames {f : f mod W == rank} of a B-frame window; three iterations on given
    pixel draws and jitter (window_graph: replayed from the segmented capture); saves the model, this rank's poses and the gathered window"""
    import torch.distributed as dist
    import test_gpu_window as TW
    from unislam_amd.dist import broadcast_parameters, shard_frames
    B, P, n_per = 6, 300, 40
    extra = (4, 15) if variant == "window_extra" else None
    c2ws, depths, colors, dirs = TW._window(B, P, 31)
    dec, es, ec = T._scene(us, False, seed=11)
    own = shard_frames(B, rank, world)
    step = us.MapStep(es, ec, dec, T.BOUND, 32, 8, 0.06, T.W, T.LR, max_rays=len(own) * n_per + (60 if extra else 0), group=True,
                      sharded_adam=(variant == "window_sharded"))
    broadcast_parameters(step.flat)
    win = us.MapWindow.sharded(step, c2ws, depths, colors, dirs, n_per, joint_opt=True, cam_lr=1e-3, extra=extra, has_zero_depth=False)
    assert win.b == len(own) and win.first == (1 if rank == 0 else 0)
    sel = torch.tensor(own)
    graph = variant == "window_graph"
    if graph:
        win.capture(t_rand=True, device_draw=False)
    losses = []
    for it in range(3):
        idx, idx2, tr, tr2 = window_draws(B, P, n_per, extra, 40, it)
        ia = idx[sel].to(T.DEV)
        t_rand = tr[sel].reshape(-1, 40)
        ib = None
        if win.extra:
            newest = [f for f in own if f >= B - extra[0]]
            rows = torch.tensor([f - (B - extra[0]) for f in newest])
            ib = idx2[rows].to(T.DEV)
            t_rand = torch.cat([t_rand, tr2[rows].reshape(-1, 40)])
        if graph:
            win.t_rand.copy_(t_rand.to(T.DEV))
            losses.append(float(win.replay(ia, ib)))
        else:
            losses.append(float(win.iterate(ia, ib, t_rand=t_rand.to(T.DEV))))
    torch.cuda.synchronize()
    torch.save({"flat": step.flat.detach().cpu(), "losses": losses, "step_dev": float(step.step_dev[0]), "c2ws": win.c2ws().cpu(),
                "c2ws_all": win.c2ws_all().cpu(), "segments": len(win._graph.segments) if graph else 0}, f"{out}.{rank}")


if __name__ == "__main__":
    main()
