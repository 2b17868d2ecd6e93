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

# Every captured graph is kept alive here for as long as ANY captured graph may still be replayed.  Reason (r5, reproduced twice each way on
# the MI355X box with `pytest tests/test_gpu_dropin.py tests/test_gpu_slam.py`; native backtrace in profiles/r05_hipgraph_destroy_segv.txt):
# with the HIP runtime torch 2.10+rocm7.0 bundles, destroying ONE graph that has parallel branches (hipGraphExecDestroy, reached from the
# garbage collector when a MapStep / window / SLAM object dies) leaves other multi-branch graph execs with dangling branch streams: a later
# hipGraphLaunch dies in hip::Graph::UpdateStreams (SIGSEGV on the host).  All graphs of this package fork side streams.
# r6, first form: a FIFO (destroy the OLDEST graph once its owner is dead, on the reading that only OLDER graphs break).  That reading was
# wrong: tools/graph_fifo_check.py ran clean, but `pytest tests/test_gpu_window.py` on its own crashed the same way after two older graphs
# were destroyed behind a living newer one and a further graph was captured (profiles/r06_graph_release_rules.txt) -- a use-after-free that
# only shows when the freed memory is reused.  r6, final form ("idle"): each entry holds the graph and a weak reference to the object that
# replays it (CapturedIteration / SegmentedGraph); before a capture the registry is emptied ONLY IF every owner in it is dead -- no living
# graph exec is left to hold a dangling stream.  A process that keeps one long-lived graph (the tracker's) and re-captures others (a window
# per mapped frame, a re-capture after KeyframeArena.grow()) therefore holds every graph it captured: capture one graph per KIND of
# iteration and rebind its inputs (ArenaWindow.bind does); the registry warns beyond 256.  release_all() drops everything; only safe when no
# captured graph will be replayed again.  US_GRAPH_RELEASE=never|idle|fifo picks the rule ("fifo" only to reproduce the fault).
_KEEP = []                  # [(graph, weakref to its owner | None)], oldest first
_RELEASE_DEFAULT = "idle"
_WARN_AT = 256              # a registry this long means per-frame captures behind a long-lived graph: say so once
_warned = [False]


def _keep(g, owner=None):
    if os.environ.get("US_KEEP_GRAPHS", "1") != "0":     # "0": the r4 behaviour (for reproducing the runtime fault)
        _KEEP.append((g, weakref.ref(owner) if owner is not None else None))
        if len(_KEEP) > _WARN_AT and not _warned[0]:
            _warned[0] = True
            warnings.warn(f"unislam_amd.graph: {len(_KEEP)} captured hipGraphs are alive (each holds its launch records and a private memory pool). "
                          "Capture one graph per KIND of iteration and rebind its inputs (ArenaWindow.bind): graphs are only destroyed once "
                          "every object that owns one is gone.")


def collect():
    """empty the registry if EVERY owner in it is gone (see _KEEP); returns how many graphs were destroyed.  Called before every capture --
    never during one (hipGraphDestroy is not permitted while a stream captures)."""
    rule = os.environ.get("US_GRAPH_RELEASE", _RELEASE_DEFAULT)
    if rule == "never":
        return 0
    n = 0
    if rule == "fifo":                                   # the refuted first form: dead owners at the front go, oldest first
        while _KEEP and _KEEP[0][1] is not None and _KEEP[0][1]() is None:
            _KEEP.pop(0)
            n += 1
        return n
    if all(r is not None and r() is None for _, r in _KEEP):
        while _KEEP:
            _KEEP.pop(0)                                 # oldest first
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
