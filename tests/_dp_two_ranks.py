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


if __name__ == "__main__":
    main()
