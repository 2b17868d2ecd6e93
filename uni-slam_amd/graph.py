"""
hipGraph capture of a launch-bound inner loop.  The tracking iteration is ~100 small launches (ray set-up, ~30 HIP
kernels of the C ABI, the quaternion autograd graph, Adam on 7 numbers): eager it is host-bound (≈2 ms for ≈0.3 ms of
GPU work).  Every C-ABI entry point launches on torch's current stream and never synchronises or allocates, so the
whole iteration can be captured once per frame and replayed.

    it = CapturedIteration(lambda: track_step.iterate(pose, gt_color, gt_depth, n, optimizer, ...))
    for _ in range(num_cam_iters): loss, unc, valid = it.replay()      # outputs are static tensors, overwritten per replay

Requirements on `fn`: static tensor addresses (update inputs in place), no host synchronisation (no .item(), no boolean
indexing), optimisers constructed with capturable=True and already stepped at least once (warmup >= 1 does that): a
capture that contains an optimiser's lazy state initialisation would re-zero the moments on every replay.
"""
import gc
import os
import warnings
import weakref

import torch

# Every captured graph is kept alive here for as long as a NEWER one may still be replayed.  Reason (r5, reproduced twice each way on the
# MI355X box with `pytest tests/test_gpu_dropin.py tests/test_gpu_slam.py`; native backtrace in profiles/r05_hipgraph_destroy_segv.txt): with
# the HIP runtime torch 2.10+rocm7.0 bundles, destroying ONE graph that has parallel branches (hipGraphExecDestroy, reached from the garbage
# collector when a MapStep / window / SLAM object dies) leaves every OLDER multi-branch graph exec with dangling branch streams: its next
# hipGraphLaunch dies in hip::Graph::UpdateStreams (SIGSEGV on the host).  All graphs of this package fork side streams, so a graph may only
# be destroyed when no older one is alive.  r6: the registry is a FIFO -- each entry holds the graph and a weak reference to the object that
# replays it (CapturedIteration / SegmentedGraph); before every new capture the entries at the FRONT whose owners are dead are dropped,
# oldest first (the oldest graph has no older one to break).  So a long run that re-captures (a window per mapped frame, a re-capture after
# KeyframeArena.grow() or a MapStep reallocation) holds the graphs that are still in use plus those younger than the oldest one in use, not
# every graph ever captured.  release_all() drops everything; only safe when no captured graph will be replayed again.
_KEEP = []                  # [(graph, weakref to its owner | None)], oldest first
_WARN_AT = 256              # a registry this long means per-frame captures whose owners stay alive: say so once
_warned = [False]


def _keep(g, owner=None):
    if os.environ.get("US_KEEP_GRAPHS", "1") != "0":     # "0": the old behaviour (for reproducing the runtime fault)
        _KEEP.append((g, weakref.ref(owner) if owner is not None else None))
        if len(_KEEP) > _WARN_AT and not _warned[0]:
            _warned[0] = True
            warnings.warn(f"unislam_amd.graph: {len(_KEEP)} captured hipGraphs are alive (each holds its launch records and a private memory pool). "
                          "Capture one graph per KIND of iteration and rebind its inputs (ArenaWindow.bind), or drop the objects that own old "
                          "graphs: they are destroyed oldest-first at the next capture.")


def collect():
    """destroy the graphs at the FRONT of the registry whose owners are gone (oldest first: safe, see _KEEP); returns how many.  Called
    before every capture -- never during one (hipGraphDestroy is not permitted while a stream captures)."""
    n = 0
    while _KEEP and _KEEP[0][1] is not None and _KEEP[0][1]() is None:
        _KEEP.pop(0)
        n += 1
    return n


def release_all():
    """destroy every graph captured so far.  Only safe when none of them will be replayed again (see _KEEP)."""
    n = len(_KEEP)
    del _KEEP[:]
    gc.collect()
    return n


class CapturedIteration:
    def __init__(self, fn, warmup=3):
        self.fn = fn
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                      # warm-up on a side stream (allocator + lazy init), as torch requires
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        collect()
        self.graph = torch.cuda.CUDAGraph()
        _keep(self.graph, self)
        # No cyclic garbage collection while the stream captures: an older graph that is only reachable from a garbage cycle would be
        # destroyed by the collector in the middle of the capture, and hipGraphDestroy is "not permitted when stream is capturing"
        # (seen in a 40-frame SLAM run with one captured MapWindow per mapped frame).  The callable is dropped afterwards: it usually
        # closes over the object that owns this one, and that cycle is what kept old graphs alive until a collection.
        was = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                self.out = fn()
        finally:
            if was:
                gc.enable()
        self.fn = None

    def replay(self):
        self.graph.replay()
        return self.out


class SegmentedGraph:
    """
    One iteration as SEVERAL hipGraphs with eager operations between them -- the data-parallel mapping step: its rank-local launches are
    captured, the collectives (torch.distributed work objects on RCCL's own stream) stay eager.

        sg = SegmentedGraph(lambda cut: step(cut), join=step._join_side_streams)
        sg.replay()

    `fn(cut)` is run once under capture; wherever it calls cut(op), the current graph ends (after join(): side streams the caller forked
    must have re-joined the capturing stream, a capture cannot end with forked work outstanding), `op` is noted, and a new graph begins.
    replay() launches graph, op(), graph, op(), ... on the current stream and returns what fn returned (static tensors).
    The capture is thread-local: a process group's watchdog thread may query its events meanwhile.
    """

    def __init__(self, fn, join=None):
        self.segments = []
        collect()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        was = gc.isenabled()
        gc.disable()                                    # (see CapturedIteration: no graph may be destroyed while a stream captures)
        cur = [None]

        def begin():
            cur[0] = torch.cuda.CUDAGraph()
            _keep(cur[0], self)
            cur[0].capture_begin(capture_error_mode="thread_local")

        def end(op):
            if join is not None:
                join()
            # a segment that recorded NOTHING (two cuts in a row: e.g. the optimiser in parts with nothing between the collectives) is not
            # launched on replay -- an empty hipGraph still costs a graph launch (torch says "The CUDA Graph is empty" at capture_end)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                cur[0].capture_end()
            empty = any("Graph is empty" in str(x.message) for x in w)
            self.segments.append((None if empty else cur[0], op))

        def cut(op):
            end(op)
            begin()

        try:
            with torch.cuda.stream(s):
                begin()
                try:
                    self.out = fn(cut)
                except BaseException:
                    # fn failed under capture: close the open capture so that the stream is usable again, but let the ORIGINAL error
                    # through (ending an invalidated capture usually raises too) and keep no half-built segments
                    try:
                        end(None)
                    except Exception:
                        pass
                    self.segments = []
                    raise
                end(None)
        finally:
            if was:
                gc.enable()
        torch.cuda.current_stream().wait_stream(s)

    def replay(self):
        for g, op in self.segments:
            if g is not None:
                g.replay()
            if op is not None:
                op()
        return self.out

    @property
    def n_launched(self):
        """segments that hold work (the others are skipped on replay)"""
        return sum(1 for g, _ in self.segments if g is not None)
