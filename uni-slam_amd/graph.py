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

import torch


class CapturedIteration:
    def __init__(self, fn, warmup=3):
        self.fn = fn
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                      # warm-up on a side stream (allocator + lazy init), as torch requires
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
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
