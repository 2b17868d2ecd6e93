"""
Data-parallel mapping over the GPUs of one node (BASELINE.json config 4): every rank renders its own slice of the
rays (frames shard naturally), the model is replicated, and ONE all-reduce of the flat gradient buffer
per optimiser step keeps the replicas identical.  `backend="nccl"` is RCCL over xGMI on ROCm.

Parity subtlety: every loss term of the reference is a mean over a DATA-DEPENDENT number of elements
(src/Mapper.py:167-173,427-430).  Averaging per-rank means is not the single-process loss.  So the engine exposes its
local (sum, count) statistics after the forward pass; they are all-reduced first (40 bytes), the backward pass
scales by the GLOBAL counts, and the gradient all-reduce is then a plain SUM.  With that, N ranks on N slices produce
the gradient of one process on the concatenated batch (tests/test_dist_gloo.py, world_size 2 on CPU).

The engine protocol (MapStep implements it on the HIP kernels; the CPU test drives the same function with the oracle):
    engine.forward(*batch) -> fills engine.stats  (tensor[10]: 5 sums, 5 counts of the LOCAL rays)
    engine.backward(on_ready=None) -> fills engine.grad (flat tensor) using engine.stats; returns the loss tensor;
                           may call on_ready(view_of_grad) whenever a contiguous segment is final (enables overlap)
    engine.adam_step()     -> applies the (reduced) gradient
"""
import os

import torch
import torch.distributed as dist


def _pg(group):
    return None if group is True else group


def dp_iterate(engine, batch, group=None):
    """
    One optimiser step. group: None (single process) | True (default process group) | a process group.

    Overlap: engine.backward(on_ready) calls on_ready(view) as soon as a contiguous segment of engine.grad is final; each
    segment's all-reduce is issued asynchronously right then (RCCL runs it on its own stream, ordered after the kernels
    launched so far) and overlaps with the rest of the backward pass.  MapStep finishes the 44.7 MB colour-table segment
    first, so its reduction hides behind the SDF decoder + SDF table backward; all segments are waited for before Adam.
    """
    engine.forward(*batch)
    if group is not None:
        dist.all_reduce(engine.stats, op=dist.ReduceOp.SUM, group=_pg(group))
    works = []

    def on_ready(view):
        if group is not None:
            works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=_pg(group), async_op=True))

    loss = engine.backward(on_ready)
    if group is not None:
        if not works:                                   # an engine that does not announce segments: one reduction at the end
            dist.all_reduce(engine.grad, op=dist.ReduceOp.SUM, group=_pg(group))
        for w in works:
            w.wait()
    engine.adam_step()
    return loss


def init_from_env(backend=None):
    """
    torchrun-style initialisation (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment),
    one process per GPU.  Returns (rank, local_rank, world_size).
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_frames(n_frames, rank, world):
    """frames {f : f mod world == rank}"""
    return list(range(rank, n_frames, world))


def broadcast_parameters(flat, group=True, src=0):
    """make the replicas bit-identical before the first step"""
    dist.broadcast(flat, src=src, group=_pg(group))
