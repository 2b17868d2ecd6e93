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


class GradComm:
    """
    The collectives of one data-parallel optimiser step, as operations that can be issued eagerly or between the hipGraph segments of a
    captured step (graph.SegmentedGraph): stats() -- the 40-byte all-reduce of the loss sums / counts; announce(view) -> op -- notes a
    finished contiguous segment of engine.grad and returns the operation that starts its reduction (asynchronously: RCCL runs it on
    its own stream, ordered after the kernels queued so far); finish() -- waits for all of them (stream-level for RCCL) and widens a
    bf16 payload again; sharded: ranges() for engine.adam_step and gather() for the updated parameter slices.

    payload "fp32": the gradient travels as it is (N ranks reproduce one process on the concatenated batch).  "bf16": every announced
    segment is rounded to bfloat16 for the reduction (half the xGMI bytes; not bit-faithful to one process); "bf16_colour": only the
    colour table's segment is.  sharded: each segment is
    reduce-scattered in place, Adam runs on the 1/W slices this rank owns, the parameter slices are all-gathered.
    """

    def __init__(self, engine, group, grad_comm=None, sharded=None):
        self.engine, self.group, self.pg = engine, group, _pg(group)
        if grad_comm is None:
            grad_comm = getattr(engine, "grad_comm", None)
        if grad_comm not in (None, "fp32", "bf16", "bf16_colour", torch.bfloat16, torch.float32):
            raise ValueError(f"dp_iterate: grad_comm {grad_comm!r} not in (None, 'fp32', 'bf16', 'bf16_colour')")
        self.narrow = grad_comm in ("bf16", "bf16_colour", torch.bfloat16)
        # "bf16_colour": only segments inside the colour table travel as bfloat16 (44.7 of the 51.7 MB of room0; they feed the colour
        # decoder alone -- the geometry, i.e. sdf table, decoders and beta, keeps its fp32 sum over the ranks)
        self.narrow_from = int(getattr(engine, "o_tab_c", 0)) if grad_comm == "bf16_colour" else 0
        self.sharded = bool(getattr(engine, "sharded_adam", False) if sharded is None else sharded)
        if self.sharded and self.narrow:                    # refused before any work is queued
            raise ValueError("dp_iterate: a bf16 gradient payload and sharded Adam are exclusive (the reduce-scatter works in place "
                             "on the fp32 gradient buffer)")
        self.W, self.r = dist.get_world_size(self.pg), dist.get_rank(self.pg)
        self.nccl = dist.get_backend(self.pg) == "nccl"
        self.segments = []                                  # (lo, n) of the announced segments, in announcement order
        self.works = []
        # direct: the engine's optimiser reads narrow segments from the bfloat16 image itself (engine._grad_bf16_from: the first index that
        # lives there), so finish() does not widen them back.  Decided when a segment is ANNOUNCED, i.e. also while a step is being captured.
        self.direct = self.narrow and hasattr(engine, "_grad_bf16_from")
        if hasattr(engine, "_grad_bf16_from"):
            engine._grad_bf16_from = None
        # in_kernel: the engine's accumulate pass leaves the colour table's gradient as a bfloat16 image itself (MapStep: us_hashgrid_bwd_joint_img),
        # so no narrowing pass runs over that segment.  The image is allocated HERE, before any launch (and any capture) takes its address.
        self.in_kernel = False
        if self.direct and grad_comm == "bf16_colour" and hasattr(engine, "enable_grad_image"):
            img = getattr(engine, "_grad_bf16", None)
            if img is None or img.shape != engine.grad.shape or img.device != engine.grad.device:
                engine._grad_bf16 = torch.empty_like(engine.grad, dtype=torch.bfloat16)
            self.in_kernel = bool(engine.enable_grad_image(True))
        elif hasattr(engine, "enable_grad_image"):
            engine.enable_grad_image(False)

    # -- operations (each one is also a valid `op` of SegmentedGraph.cut)
    def stats(self):
        self.works = []
        dist.all_reduce(self.engine.stats, op=dist.ReduceOp.SUM, group=self.pg)

    def announce(self, view):
        e = self.engine
        n = view.numel()
        lo = (view.data_ptr() - e.grad.data_ptr()) // view.element_size()      # index of the segment in the flat buffers
        if self.sharded and n % self.W:
            raise ValueError(f"sharded Adam: a gradient segment of {n} elements does not split over {self.W} ranks")
        self.segments.append((lo, n))
        if self.sharded:
            sz = n // self.W
            mine = view[self.r * sz:(self.r + 1) * sz]

            def op():
                if self.nccl:                               # in place: slice r of the input
                    w = dist.reduce_scatter_tensor(mine, view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                else:                                       # gloo has no reduce-scatter: the all-reduce leaves the same values in the slice
                    w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self.works.append((w, None, None))
            return op
        buf = view
        if self.narrow and lo >= self.narrow_from:          # one bf16 image of the gradient buffer, allocated once: its slices are the payloads
            img = getattr(e, "_grad_bf16", None)
            if img is None or img.shape != e.grad.shape or img.device != e.grad.device:
                img = e._grad_bf16 = torch.empty_like(e.grad, dtype=torch.bfloat16)
            buf = img[lo:lo + n]
            if self.direct:
                e._grad_bf16_from = lo if e._grad_bf16_from is None else min(e._grad_bf16_from, lo)

        written = self.in_kernel and lo == self.narrow_from and getattr(e, "_grad_image_written", False)      # (the colour table's segment)

        def op():
            if buf is not view and not written:
                buf.copy_(view)
            self.works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), view, buf))
        return op

    def finish(self, first=None):
        """wait for the announced segments (stream-level over RCCL); first = k: only the first k of them (announcement order), the
        others stay in flight.  (An engine whose optimiser reads the bfloat16 image itself -- self.direct -- needs no widening.)"""
        n = len(self.works) if first is None else min(int(first), len(self.works))
        for w, view, buf in self.works[:n]:
            w.wait()
            if buf is not None and buf is not view and not self.direct:
                view.copy_(buf)
        self.works = self.works[n:]

    def finish_first(self):
        self.finish(first=1)

    def ranges(self):
        return [(lo + self.r * (n // self.W), lo + (self.r + 1) * (n // self.W)) for lo, n in self.segments]

    def gather(self):
        e, W, r = self.engine, self.W, self.r
        ws = []
        for lo, n in self.segments:
            sz = n // W
            whole = e.flat[lo:lo + n]
            if self.nccl:                                   # in place
                ws.append(dist.all_gather_into_tensor(whole, whole[r * sz:(r + 1) * sz], group=self.pg, async_op=True))
            else:
                ws.append(dist.all_gather([whole[k * sz:(k + 1) * sz] for k in range(W)], whole[r * sz:(r + 1) * sz].clone(), group=self.pg,
                                          async_op=True))
        for w in ws:
            w.wait()


def dp_iterate(engine, batch, group=None, grad_comm=None, ray_grads=False, before_adam=None, cut=None, comm=None):
    """
    One optimiser step. group: None (single process) | True (default process group) | a process group.
    grad_comm: None / "fp32" -> the gradient travels as it is; "bf16" -> every announced segment is rounded to bfloat16 for
    the all-reduce and widened again before Adam (half the bytes on xGMI; the sum over ranks then carries bf16 rounding, so
    N ranks no longer reproduce one process bit for bit -- opt-in, for when the all-reduce bounds the step).
    ray_grads: engine.backward also forms dL/d(rays) (the joint pose optimisation of src/Mapper.py:359-376); before_adam(): the caller's
    rank-local work between the backward pass and the optimiser (MapWindow: the pose step of the frames this rank owns -- a frame's rays
    live on ONE rank, so its pose gradient needs no reduction), queued while the gradient segments travel.
    cut: None -> every collective is issued where it stands (eager); a SegmentedGraph's cut -> the rank-local launches are being
    captured, each collective ends a graph segment and runs between the segments at replay time.

    Overlap: engine.backward(on_ready) calls on_ready(view) as soon as a contiguous segment of engine.grad is final; each
    segment's all-reduce is issued asynchronously right then (RCCL runs it on its own stream, ordered after the kernels
    launched so far) and overlaps with the rest of the backward pass.  MapStep finishes the 44.7 MB colour-table segment
    first, so its reduction hides behind the SDF table's share of the backward pass; all segments are waited for before Adam.
    """
    if group is None:
        engine.forward(*batch)
        loss = engine.backward(**({"ray_grads": True} if ray_grads else {}))
        if before_adam is not None:
            before_adam()
        engine.adam_step()
        return loss
    if comm is None:
        comm = GradComm(engine, group, grad_comm)
    comm.segments = []
    run = (lambda op: op()) if cut is None else cut
    engine.forward(*batch)
    run(comm.stats)
    loss = engine.backward(lambda view: run(comm.announce(view)), **({"ray_grads": True} if ray_grads else {}))
    if not comm.segments:                                   # an engine that does not announce segments: one reduction at the end
        run(comm.announce(engine.grad))
    if before_adam is not None:
        before_adam()
    if comm.sharded:
        run(comm.finish)
        engine.adam_step(ranges=comm.ranges())
        run(comm.gather)
    elif getattr(engine, "adam_in_parts", False) and len(comm.segments) == 2 and comm.segments[0][0] == getattr(engine, "o_tab_c", -1):
        # the colour table's segment was announced first and is the large one: its optimiser pass (47 of 54 us) starts as soon as it has
        # arrived and covers the reduction of the small remaining segment, which only the last 7 us of the optimiser need
        run(comm.finish_first)
        engine.adam_step(part="colour")
        run(comm.finish)
        engine.adam_step(part="rest")
    else:
        run(comm.finish)
        engine.adam_step()
    return loss


def init_from_env(backend=None, timeout_s=None):
    """
    torchrun-style initialisation (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment),
    one process per GPU.  Returns (rank, local_rank, world_size).  timeout_s: the process group's collective timeout (a rank stuck in a
    collective then ends with an error after that long instead of the default 10 minutes).
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
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
