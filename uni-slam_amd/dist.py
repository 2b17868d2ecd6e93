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


def dp_iterate(engine, batch, group=None, grad_comm=None):
    """
    One optimiser step. group: None (single process) | True (default process group) | a process group.
    grad_comm: None / "fp32" -> the gradient travels as it is; "bf16" -> every announced segment is rounded to bfloat16 for
    the all-reduce and widened again before Adam (half the bytes on xGMI; the sum over ranks then carries bf16 rounding, so
    N ranks no longer reproduce one process bit for bit -- opt-in, for when the all-reduce bounds the step).

    Overlap: engine.backward(on_ready) calls on_ready(view) as soon as a contiguous segment of engine.grad is final; each
    segment's all-reduce is issued asynchronously right then (RCCL runs it on its own stream, ordered after the kernels
    launched so far) and overlaps with the rest of the backward pass.  MapStep finishes the 44.7 MB colour-table segment
    first, so its reduction hides behind the SDF decoder + SDF table backward; all segments are waited for before Adam.
    """
    if grad_comm is None:
        grad_comm = getattr(engine, "grad_comm", None)
    narrow = grad_comm in ("bf16", torch.bfloat16)
    sharded = group is not None and getattr(engine, "sharded_adam", False)
    if sharded and narrow:                              # refused before any work is queued
        raise ValueError("dp_iterate: a bf16 gradient payload and sharded Adam are exclusive (the reduce-scatter works in place "
                         "on the fp32 gradient buffer)")
    engine.forward(*batch)
    if group is not None:
        dist.all_reduce(engine.stats, op=dist.ReduceOp.SUM, group=_pg(group))
    if sharded:
        return _finish_sharded(engine, group)
    works = []

    def on_ready(view):
        if group is not None:
            buf = view
            if narrow:                                  # one bf16 image of the gradient buffer, allocated once: its slices are the payloads
                img = getattr(engine, "_grad_bf16", None)
                if img is None or img.shape != engine.grad.shape or img.device != engine.grad.device:
                    img = engine._grad_bf16 = torch.empty_like(engine.grad, dtype=torch.bfloat16)
                lo = (view.data_ptr() - engine.grad.data_ptr()) // view.element_size()
                buf = img[lo:lo + view.numel()]
                buf.copy_(view)
            works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=_pg(group), async_op=True), view, buf))

    loss = engine.backward(on_ready if group is not None else None)     # (a single process has no use for per-segment readiness)
    if group is not None:
        if not works:                                   # an engine that does not announce segments: one reduction at the end
            on_ready(engine.grad)
        for w, view, buf in works:
            w.wait()
            if buf is not view:
                view.copy_(buf)
    engine.adam_step()
    return loss


def _finish_sharded(engine, group):
    """
    The same step with the optimiser sharded over the ranks ("reduce-scatter + sharded Adam + all-gather"): every
    announced gradient segment is reduce-scattered (rank r receives the sum of slice r), Adam runs on the slices this rank owns
    (1/W of the 7 x 51.7 MB of optimiser traffic, 1/W of the moment memory in use), and the updated parameter slices are
    all-gathered.  Same bytes on the wire as the all-reduce, replicas stay identical.  Needs engine.flat / engine.grad (flat
    buffers with the same indexing), segments whose length is a multiple of the world size, and engine.adam_step(ranges=...).
    gloo has no reduce-scatter: there (CPU tests) the segment is all-reduced, which leaves the same values in the slice.
    """
    pg = _pg(group)
    W, r = dist.get_world_size(pg), dist.get_rank(pg)
    nccl = dist.get_backend(pg) == "nccl"
    pending = []

    def on_ready(view):
        n = view.numel()
        if n % W:
            raise ValueError(f"sharded Adam: a gradient segment of {n} elements does not split over {W} ranks")
        sz = n // W
        lo = (view.data_ptr() - engine.grad.data_ptr()) // view.element_size()     # index of the segment in the flat buffers
        mine = view[r * sz:(r + 1) * sz]
        if nccl:
            w = dist.reduce_scatter_tensor(mine, view, op=dist.ReduceOp.SUM, group=pg, async_op=True)   # in place: slice r of the input
        else:
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=pg, async_op=True)
        pending.append((w, lo, n, sz))

    loss = engine.backward(on_ready)
    if not pending:
        on_ready(engine.grad)
    for w, _, _, _ in pending:
        w.wait()
    engine.adam_step(ranges=[(lo + r * sz, lo + (r + 1) * sz) for _, lo, _, sz in pending])
    gathers = []
    for _, lo, n, sz in pending:
        whole = engine.flat[lo:lo + n]
        if nccl:
            gathers.append(dist.all_gather_into_tensor(whole, whole[r * sz:(r + 1) * sz], group=pg, async_op=True))   # in place
        else:
            gathers.append(dist.all_gather([whole[k * sz:(k + 1) * sz] for k in range(W)], whole[r * sz:(r + 1) * sz].clone(), group=pg,
                                           async_op=True))
    for w in gathers:
        w.wait()
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


def all_agree(flag, group=True, device="cpu"):
    """True on every rank iff `flag` is true on EVERY rank (one MIN all-reduce); without a process group: flag itself.
    For loops whose length a rank would otherwise decide by itself -- a wall-clock budget, a host-side convergence test -- when the
    loop body contains collectives: ranks that leave such a loop after different numbers of turns wait for each other forever
    (bench.py's set-up phase did)."""
    if group is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size(_pg(group)) == 1:
        return bool(flag)
    f = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=_pg(group))
    return bool(f.item() > 0.5)


def shard_frames(n_frames, rank, world):
    """frames {f : f mod world == rank}"""
    return list(range(rank, n_frames, world))


def broadcast_parameters(flat, group=True, src=0):
    """make the replicas bit-identical before the first step"""
    dist.broadcast(flat, src=src, group=_pg(group))
