"""
MapWindow -- Mapper.optimize_mapping's loop (src/Mapper.py:366-445) for one mapped frame with the window's camera poses ON THE
DEVICE: the reference's default `joint_opt: True` (configs/UNISLAM.yaml:50), which is on from the fifth keyframe (src/Mapper.py:519),
i.e. for nearly the whole sequence.  Per iteration the reference runs

    c2ws_ = cat(c2ws[0:1], cam_pose_to_matrix(cam_poses))                      src/Mapper.py:372-376, src/common.py:196-208
    rays  = get_samples_all(..., c2ws_, pools)  [+ 10 x 200 extra rays]        src/Mapper.py:379-393, src/common.py:152-166
    filter, render, loss, backward                                             src/Mapper.py:396-444
    optimizer.step()   (decoders, tables, AND the poses as a fourth group)     src/Mapper.py:359-364,445

through autograd.  Here the same iteration is a fixed sequence of HIP launches on static buffers -- us_window_rays (quaternion ->
rotation -> gather + rotate), MapStep.forward / backward (+ us_hashgrid_bwd_input_rays: both grids' input gradient reduced to the
rays in one launch), us_pose_window_step (per-frame pose gradient + Adam), MapStep.adam_step -- with no torch autograd and no host
synchronisation, so it can be captured into ONE hipGraph (capture() / replay(), pixel draw included).
With joint_opt off the window is the same machine without the pose step (the poses then never move).
Data-parallel (a process group on the MapStep): MapWindow.sharded() gives every rank its share of the window's frames; a frame's rays
live on one rank, so its pose and the pose's Adam state do too -- the iteration exchanges loss statistics and model gradients only, and
c2ws_all() collects the poses when the window is done.
"""
import ctypes

import torch

from . import _lib as L
from .common import cam_pose_to_matrix, matrix_to_cam_pose


class MapWindow:
    def __init__(self, step, c2ws, depths, colors, dirs, n_per_frame, joint_opt=True, cam_lr=1e-3, extra=None, has_zero_depth=None,
                 fixed_first=True):
        """
        step: MapStep (owns the model, the optimiser state and the render buffers; call step.reset_optimizer() first, as
              Mapper.optimize_mapping builds a fresh Adam per mapped frame, src/Mapper.py:358-364);
        c2ws [b,4,4]: the window's poses, the OLDEST first -- it stays fixed (src/Mapper.py:374); depths [b,P], colors [b,P,3],
        dirs [b,P,3]: the frames' pixel pools (camera-frame directions); n_per_frame = mapping_pixels // b (src/Mapper.py:315);
        extra: None | (n_frames, n_pixels): n_pixels more rays from each of the newest n_frames frames (src/Mapper.py:385-393: 10 x 200
               once the keyframe list has more than 20 entries and the tracker is not tracking back);
        fixed_first: frame 0 is the window's oldest frame and keeps its pose (src/Mapper.py:374).  False: every frame given here is
               optimised -- a data-parallel rank whose share of the window does not hold the oldest frame (MapWindow.sharded);
        cam_lr: cfg['mapping']['joint_opt_cam_lr'] (src/Mapper.py:362); has_zero_depth: False -> every pool pixel carries a depth (the
               zero-depth branch's launches are skipped); None / True -> the branch of src/utils/Renderer.py:104-130 runs with its row
               count on the device (no host synchronisation: the caller need not look at the pools).
        """
        self.step = step
        dev = step.device
        b, P = depths.shape
        if b < 1 or c2ws.shape != (b, 4, 4) or colors.shape != (b, P, 3) or dirs.shape != (b, P, 3):
            raise L.UniSlamHipError("MapWindow: c2ws [b,4,4], depths [b,P], colors [b,P,3], dirs [b,P,3] expected")
        self.b, self.P, self.n_per = b, P, int(n_per_frame)
        self.first = 1 if fixed_first else 0                       # frames [first, b) are optimised: pose j belongs to frame j + first
        self.joint_opt = bool(joint_opt) and b > self.first
        self.cam_lr = float(cam_lr)
        self.pool_d, self.pool_c, self.pool_r = L.f32(depths.to(dev)), L.f32(colors.to(dev)), L.f32(dirs.to(dev))
        c2ws = L.f32(c2ws.detach().to(dev))
        self.c2w_first = c2ws[0].clone() if fixed_first else None
        n_p = max(b - self.first, 1)
        # (With joint_opt off the reference renders from the matrices as given, src/Mapper.py:377-378; here frames first.. always pass
        #  through quaternion + translation, src/common.py:182-208 -- an fp32 round trip of ~1e-7 that also re-orthonormalises a pose
        #  that is not exactly a rotation.)
        self.poses = matrix_to_cam_pose(c2ws[self.first:]).contiguous() if b > self.first else torch.zeros((1, 7), device=dev)   # src/Mapper.py:360
        self.pm, self.pv = torch.zeros((n_p, 7), device=dev), torch.zeros((n_p, 7), device=dev)
        self.g_pose = torch.zeros((n_p, 7), device=dev)
        if extra is not None and (extra[0] <= 0 or extra[1] <= 0):
            extra = None
        self.extra = None if extra is None else (min(int(extra[0]), b), int(extra[1]))
        self.R_a = b * self.n_per
        self.R = self.R_a + (self.extra[0] * self.extra[1] if self.extra else 0)
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.ro, self.rd, self.gd, self.gc, self.dirs = f(self.R, 3), f(self.R, 3), f(self.R), f(self.R, 3), f(self.R, 3)
        self.idx_a = torch.zeros((b, self.n_per), dtype=torch.int64, device=dev)
        self.idx_b = torch.zeros(self.extra, dtype=torch.int64, device=dev) if self.extra else None
        self.has_zero = True if has_zero_depth is None else bool(has_zero_depth)
        self._graph, self.t_rand, self.zd_draws = None, None, None
        self._comm, self._shard = None, None                      # data-parallel: dist.GradComm of the step; (own frames, B, group) of sharded()
        if self.R > step.max_rays:
            step._alloc(self.R)

    # ------------------------------------------------------------------------------------------ one iteration
    def draw(self, indices=None, indices_extra=None):
        """the pixel draw of common.get_samples_all (src/common.py:155) into the static index tensors (torch.randint, or given indices)"""
        if indices is None:
            torch.randint(self.P, (self.b, self.n_per), device=self.idx_a.device, out=self.idx_a)
        else:
            self.idx_a.copy_(indices.reshape(self.b, self.n_per))
        if self.extra:
            if indices_extra is None:
                torch.randint(self.P, self.extra, device=self.idx_b.device, out=self.idx_b)
            else:
                self.idx_b.copy_(indices_extra.reshape(self.extra))

    def rays(self):
        """(rays_o, rays_d, gt_depth, gt_color) of the drawn pixels (self.idx_a / idx_b) at the current poses: us_window_rays alone"""
        lib, st, P, b = L.lib(), L.stream(), L.ptr, self.b
        poses = P(self.poses) if b > self.first else None
        L.check(lib.us_window_rays(P(self.c2w_first), poses, P(self.pool_d), P(self.pool_c), P(self.pool_r), P(self.idx_a), self.P, 0, b,
                                   self.n_per, P(self.ro), P(self.rd), P(self.gd), P(self.gc), P(self.dirs), st), "us_window_rays")
        if self.extra:
            nf, ne = self.extra
            r0 = self.R_a
            L.check(lib.us_window_rays(P(self.c2w_first), poses, P(self.pool_d), P(self.pool_c), P(self.pool_r), P(self.idx_b), self.P, b - nf, nf,
                                       ne, self._off(self.ro, r0, 3), self._off(self.rd, r0, 3), self._off(self.gd, r0, 1), self._off(self.gc, r0, 3),
                                       self._off(self.dirs, r0, 3), st), "us_window_rays")
        return self.ro, self.rd, self.gd, self.gc

    def _off(self, t, rows, width):
        return ctypes.c_void_p(t.data_ptr() + 4 * rows * width)

    def _launches(self, t_rand=None, zero_depth_draws=None, device_draw=False, cut=None):
        """everything after the draw: rays from the current poses + samples (ONE launch: us_window_sample; with device_draw it also
        draws the pixels, like the jitter from a counter-based generator salted with the device-side step count), render + loss +
        backward, pose step, Adam.  With a process group on the MapStep the same launches run as a data-parallel step (dist.dp_iterate):
        loss statistics and gradient segments are reduced over the ranks, the pose step stays rank-local (a frame's rays live on one
        rank); cut: the SegmentedGraph hook of capture()."""
        P, s = L.ptr, self.step
        if t_rand is None:
            t_rand = self.t_rand                                 # the static jitter tensor of capture(t_rand=True), if any
        if zero_depth_draws is None:
            zero_depth_draws = self.zd_draws                     # ... and the static draws of the zero-depth branch
        if self.R > s.max_rays:
            s._alloc(self.R)
        s.rng_calls += 1
        self._sample(P(L.f32(t_rand)) if (s.perturb and t_rand is not None) else None, device_draw)
        if s.group is not None:
            from .dist import dp_iterate, GradComm
            if self._comm is None or self._comm.engine is not s:
                self._comm = GradComm(s, s.group)
            s.store_dydx = self.joint_opt                        # the encoder leaves dy/dx for the pose gradient (no second gather pass)
            try:
                return dp_iterate(s, (self.ro, self.rd, self.gd, self.gc, t_rand, self.has_zero, zero_depth_draws, True, True), s.group,
                                  ray_grads=self.joint_opt, before_adam=self._pose_step if self.joint_opt else None, cut=cut, comm=self._comm)
            finally:
                s.store_dydx = False
        if not self.joint_opt:
            return s.iterate(self.ro, self.rd, self.gd, self.gc, t_rand=t_rand, has_zero_depth=self.has_zero, presampled=True,
                             zero_depth_draws=zero_depth_draws)
        s.store_dydx = True
        try:
            s.forward(self.ro, self.rd, self.gd, self.gc, t_rand, self.has_zero, zero_depth_draws, presampled=True)
        finally:
            s.store_dydx = False
        loss = s.backward(ray_grads=True, fold=True)              # (adam_step() below sums the decoder-gradient partials itself)
        if s.one_launch_adam and getattr(s, "_folded", False):
            # the poses' group rides in the model's optimiser launch (us_adam_step_model): one workgroup per frame ahead of the tables' pass
            # instead of a launch of its own (us_pose_window_step / us_arena_pose_step: same arithmetic)
            s.adam_step(poses=self._pose_desc())
        else:
            self._pose_step()
            s.adam_step()
        return loss

    def _sample(self, tr, device_draw):
        """poses -> pool pixels -> rays -> pre-filter flag, sorted + jittered z, unit-cube points: ONE launch (us_window_sample)"""
        lib, st, P, s, b = L.lib(), L.stream(), L.ptr, self.step, self.b
        poses = P(self.poses) if b > self.first else None
        nf, ne = self.extra if self.extra else (0, 0)
        ia, ib = (None, None) if device_draw else (P(self.idx_a), P(self.idx_b) if self.extra else None)
        L.check(lib.us_window_sample(P(self.c2w_first), poses, b, self.n_per, nf, ne, P(self.pool_d), P(self.pool_c), P(self.pool_r), self.P, ia, ib,
                                     s.bhost, P(s.t_uni), s.n_strat, P(s.t_surf), s.n_imp, ctypes.c_float(1.2), ctypes.c_float(1.5 * s.truncation),
                                     ctypes.c_float(3 * s.truncation), tr, s._seed(0), P(s.step_dev), 1 if s.perturb else 0, P(self.ro),
                                     P(self.rd), P(self.dirs), P(self.gd), P(self.gc), P(s.valid), P(s.z), P(s.pts), st), "us_window_sample")

    def probe_valid(self):
        """the sampling launch ALONE on the pixels of draw() at the current poses: (pre-filter flags [R] uint8, gt_depth [R]) as the next
        iterate() will see them (padding rows of an ArenaWindow come out invalid).  For callers that replay a recorded random stream
        (slam.TorchDraws): the reference draws its jitter for the rays that passed the pre-filter only (src/Mapper.py:396-406 compacts,
        src/utils/Renderer.py:55 draws [R', S])."""
        self._sample(None, False)
        return self.step.valid[:self.R], self.gd

    def _pose_desc(self):
        """_pose_step()'s arguments as the descriptor us_adam_step_model takes"""
        P, s = (lambda t: ctypes.c_void_p(t.data_ptr())), self.step
        b, first = self.b, self.first
        nf, ne = self.extra if self.extra else (0, 0)
        f_b = max(b - nf, first)
        return L.PoseStepDesc(P(self.poses), b - first, P(s.g_o), P(s.g_d), P(self.dirs), first * self.n_per, self.n_per, f_b - first,
                              self.R_a + (f_b - (b - nf)) * ne, ne if nf else 0, P(self.pm), P(self.pv), P(self.g_pose), self.cam_lr, self.cam_lr,
                              None, 0)

    def _pose_step(self):
        """gradient + Adam of the poses this window optimises (one workgroup per pose), from the ray gradients the backward pass left"""
        lib, st, P, s = L.lib(), L.stream(), L.ptr, self.step
        b, first = self.b, self.first
        nf, ne = self.extra if self.extra else (0, 0)
        if not s._step_advanced:                                 # the poses are one more group of the SAME optimiser: one step count
            L.check(lib.us_adam_step_inc(P(s.step_dev), 0.9, 0.999, st), "us_adam_step_inc")
            s._step_advanced = True
        lr = self.cam_lr                                         # the pose group is appended with its plain lr (src/Mapper.py:362): no lr_factor
        # pose j = window frame j + first: rows [(j + first) n_per, (j + first + 1) n_per) of the first block, and -- for the newest nf
        # frames, b - nf .. b - 1 -- ne rows each of the extra block behind it
        f_b = max(b - nf, first)                                 # the first optimised frame that owns rows of the extra block
        L.check(lib.us_pose_window_step(P(self.poses), b - first, P(s.g_o), P(s.g_d), P(self.dirs), first * self.n_per, self.n_per, f_b - first,
                                        self.R_a + (f_b - (b - nf)) * ne, ne if nf else 0, P(self.pm), P(self.pv), P(self.g_pose),
                                        lr, lr, 0.9, 0.999, 1e-8, P(s.step_dev), 0, st), "us_pose_window_step")

    def iterate(self, indices=None, indices_extra=None, t_rand=None, zero_depth_draws=None):
        """one eager iteration; returns the loss tensor [1] (device).  indices None: the pixels are drawn inside us_window_sample"""
        if indices is not None:
            self.draw(indices, indices_extra)
        return self._launches(t_rand, zero_depth_draws, device_draw=indices is None)

    # ------------------------------------------------------------------------------------------ hipGraph
    def capture(self, t_rand=False, device_draw=True, collectives=None, unroll=1):
        """capture _launches() (the zero-depth branch included: its row count stays on the device) into a hipGraph: replay() is ONE graph
        launch, pixel draw included (device_draw; False: replay(indices) / torch.randint fill the static index tensors first).  The
        jitter comes from the in-kernel generator (varied per replay by the device-side step count) unless t_rand=True: then
        self.t_rand [R,S] (and, for a window with depth holes, self.zd_draws) are static inputs to fill.  The model, the optimiser state
        and the poses are left as they were.  With a process group on the MapStep every rank must call capture() (its warm-up iterations
        hold collectives); collectives: see below.
        unroll (single process, r5): the graph holds `unroll` CONSECUTIVE iterations and replay() runs them all (the loss of the last one
        is returned) -- between two graph launches the GPU idles for ~15 us (profiles/r05_timeline.txt), between two kernels of one graph
        for none to 6: a mapped frame's 15 iterations are three launches of a 5-iteration graph.  Every iteration of the graph draws its own
        pixels and jitter (the seeds baked into the launches differ, and the device-side step count is mixed in)."""
        from .graph import CapturedIteration, SegmentedGraph
        unroll = int(unroll)
        if unroll < 1 or (unroll > 1 and (self.step.group is not None or t_rand or not device_draw)):
            raise L.UniSlamHipError("MapWindow.capture: unroll > 1 needs a single process and the in-kernel pixel draw and jitter")
        s = self.step
        if s._step_advanced:
            raise L.UniSlamHipError("MapWindow.capture: a forward pass of the MapStep is pending; finish its optimiser step first")
        s._join_side_streams()
        self._graph = None
        self.t_rand = torch.zeros((self.R, s.S), dtype=torch.float32, device=s.device) if t_rand else None
        # (static draws of the zero-depth branch, indexed by the compacted row: jitter of the coarse pass [R, n_strat], inverse-transform draws [R, n_imp])
        self.zd_draws = (torch.zeros((self.R, s.n_strat), device=s.device), torch.zeros((self.R, s.n_imp), device=s.device)) if (t_rand and self.has_zero) else None
        # warm-up iterations would move the model: they run with all learning rates at zero, and EVERYTHING they touch is put back in the
        # `finally` below -- also when a warm-up iteration or the capture itself raises, so that a caller who catches the error can go on
        # with iterate() on an intact optimiser state
        keep = (s.flat.clone(), s.m.clone(), s.v.clone(), s.step_dev.clone(), s.opt_step, dict(s.lr), s.rng_calls, self.poses.clone(),
                self.pm.clone(), self.pv.clone(), self.cam_lr, s.probe)
        s.lr = {k: 0.0 for k in s.lr}
        self.cam_lr = 0.0
        s.probe = None
        self._device_draw = bool(device_draw)
        graph, side = None, None

        def restore():
            s.lr, self.cam_lr = keep[5], keep[10]
            s.flat.copy_(keep[0]); s.m.copy_(keep[1]); s.v.copy_(keep[2]); s.step_dev.copy_(keep[3])
            self.poses.copy_(keep[7]); self.pm.copy_(keep[8]); self.pv.copy_(keep[9])
            s.opt_step, s.rng_calls = keep[4], keep[6]
            s._dec_grad_clean = False                            # the captured backward clears the decoder gradient itself
            s._step_advanced = False
            s._folded = False

        try:
            try:
                self.draw()
                side = torch.cuda.Stream(device=s.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._launches(device_draw=self._device_draw)
            finally:
                if side is not None:
                    torch.cuda.current_stream().wait_stream(side)
                restore()                                        # the real learning rates are what the capture records
            if s.group is None:
                def body():
                    for _ in range(unroll):
                        out = self._launches(device_draw=self._device_draw)
                    return out
                graph = CapturedIteration(body, warmup=0)
            else:
                # data-parallel: the rank-local launches as hipGraph segments, the collectives between them (dist.dp_iterate's `cut`)
                # collectives "between" (default): the collectives stay eager between hipGraph segments (graph.SegmentedGraph; the form the
                # gloo world-2 / 8 tests cover).  "inside" (opt-in, RCCL only): the process group's calls are captured INTO the one graph --
                # RCCL kernels as graph nodes, one launch per step: 0.51 ms of rank-local time against 0.54 eager and 0.60 with "between" --
                # but it has only ever run on a 1-rank group (no node with more GPUs was available to the builder): work handles,
                # finish_first and the optimiser in parts inside a captured graph are untested at world > 1, a hang during replay only ends
                # at the process group's timeout, and once in ~40 runs of bench.py's 1-rank rehearsal torch's process-group watchdog thread
                # queried an event that had been recorded in the capturing stream (hipErrorCapturedEvent) and terminated the process.
                if collectives is None:
                    collectives = "between"
                if collectives not in ("inside", "between"):
                    raise L.UniSlamHipError(f"MapWindow.capture: collectives {collectives!r} not in ('inside', 'between')")
                inside = collectives == "inside"
                graph = SegmentedGraph(lambda cut: self._launches(device_draw=self._device_draw, cut=(lambda op: op()) if inside else cut),
                                       join=s._join_side_streams)
        finally:
            restore()                                            # (the capture pass does not execute; its host-side counters are undone)
            s.probe = keep[11]
        self._graph, self._graph_gen, self._unroll = graph, s.generation, unroll

    def replay(self, indices=None, indices_extra=None):
        if self._graph is None:
            raise L.UniSlamHipError("MapWindow.replay: call capture() first")
        if self.step.generation != self._graph_gen:              # e.g. a later, larger window on the same MapStep made it reallocate
            self._graph = None
            raise L.UniSlamHipError("MapWindow.replay: the MapStep's buffers were reallocated after this graph was captured (it holds the "
                                    "old addresses); capture() again")
        arena = getattr(self, "arena", None)
        if arena is not None and arena.generation != self.generation:       # KeyframeArena.grow(): the graph holds the freed pools' addresses
            self._graph = None
            raise L.UniSlamHipError("ArenaWindow.replay: the arena has grown since this graph was captured; build a new window")
        if self._device_draw:
            if indices is not None:
                raise L.UniSlamHipError("MapWindow.replay: this graph draws its pixels itself; capture(device_draw=False) to pass indices")
        else:
            self.draw(indices, indices_extra)
        self.step.opt_step += getattr(self, "_unroll", 1)
        return self._graph.replay()

    # ------------------------------------------------------------------------------------------ results
    def c2ws(self):
        """this window's poses now: [b,4,4] (a fixed first frame as given; the others from the optimised quaternion / translation,
        src/Mapper.py:449).  Of a sharded window: the frames THIS rank owns (c2ws_all() collects the whole window)."""
        opt = [cam_pose_to_matrix(self.poses)] if self.b > self.first else []
        return torch.cat(([self.c2w_first[None]] if self.first else []) + opt, dim=0).clone()

    # ------------------------------------------------------------------------------------------ data-parallel
    @classmethod
    def sharded(cls, step, c2ws, depths, colors, dirs, n_per_frame, joint_opt=True, cam_lr=1e-3, extra=None, has_zero_depth=None):
        """
        The window of a data-parallel mapping step (step.group set): every rank is handed the WHOLE window (B frames, the
        oldest first) and keeps the frames {f : f mod W == rank} -- n_per_frame rays from each, and its frames' share of the extra rays
        of the newest frames.  The rank that owns frame 0 keeps it fixed (src/Mapper.py:374).  Poses, their Adam moments and their
        gradients are rank-local for the whole loop: the iteration all-reduces the loss statistics and the model gradients, nothing else.
        """
        import torch.distributed as dist
        from .dist import shard_frames, _pg
        if step.group is None:
            raise L.UniSlamHipError("MapWindow.sharded: the MapStep has no process group")
        pg = _pg(step.group)
        W, r = dist.get_world_size(pg), dist.get_rank(pg)
        B = depths.shape[0]
        own = shard_frames(B, r, W)
        if not own:
            raise L.UniSlamHipError(f"MapWindow.sharded: {B} frames do not give rank {r} of {W} a frame")
        if extra is not None:                                    # the newest extra[0] frames of the WINDOW: those of them this rank owns
            k = sum(1 for f in own if f >= B - min(int(extra[0]), B))
            extra = (k, int(extra[1])) if k else None
        sel = torch.as_tensor(own, device=depths.device)
        pick = lambda t: t.index_select(0, sel.to(t.device))
        win = cls(step, pick(c2ws), pick(depths), pick(colors), pick(dirs), n_per_frame, joint_opt, cam_lr, extra, has_zero_depth,
                  fixed_first=(own[0] == 0))
        win._shard = (own, B, pg)
        return win

    def c2ws_all(self):
        """[B,4,4]: the poses of the whole window, every rank's frames at their window positions (one all-gather, when the loop is done:
        src/Mapper.py:447-457 writes them back to the keyframes)"""
        if self._shard is None:
            return self.c2ws()
        import torch.distributed as dist
        from .dist import shard_frames
        own, B, pg = self._shard
        W = dist.get_world_size(pg)
        n_max = -(-B // W)
        mine = torch.zeros((n_max, 4, 4), dtype=torch.float32, device=self.poses.device)
        mine[:len(own)] = self.c2ws()
        parts = [torch.empty_like(mine) for _ in range(W)]
        dist.all_gather(parts, mine, group=pg)
        out = torch.empty((B, 4, 4), dtype=torch.float32, device=mine.device)
        for k in range(W):
            fr = shard_frames(B, k, W)
            if fr:
                out[torch.as_tensor(fr, device=out.device)] = parts[k][:len(fr)]
        return out


class KeyframeArena:
    """
    Persistent device store of the keyframes' pixel pools (10 % of a frame each, src/Mapper.py:329-337,516-523): depth [K,P], color
    [K,P,3], dirs [K,P,3], one row per keyframe, written once when the keyframe is made.  The reference stacks the selected keyframes'
    pools into fresh tensors for every mapped frame (src/Mapper.py:317-356: b x 2.3 MB, and b grows with the sequence); here a mapping
    window is a list of ROW NUMBERS, and the pools never move -- 288 GB of HBM hold 500 Replica keyframes in 1.2 GB.
    Row 0 is scratch: the pool of the frame being mapped.  grow() doubles the store (new addresses: `generation` tells holders of
    captured graphs).
    """

    def __init__(self, capacity, pool_size, device):
        self.K, self.P, self.device = int(capacity), int(pool_size), device
        self.generation, self.used = 0, 1                        # row 0 = the current frame
        self._new(self.K)

    def _new(self, K):
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.device)
        self.depth, self.color, self.dirs = f(K, self.P) + 1.0, f(K, self.P, 3), f(K, self.P, 3)
        self.dirs[..., 2] = -1.0                                 # (unused rows hold a harmless pixel: straight ahead, 1 m)

    def grow(self):
        old = (self.depth, self.color, self.dirs)
        self.K *= 2
        self._new(self.K)
        for new, o in zip((self.depth, self.color, self.dirs), old):
            new[:o.shape[0]].copy_(o)
        self.generation += 1

    def alloc(self):
        if self.used >= self.K:
            self.grow()
        self.used += 1
        return self.used - 1

    def put(self, row, color, depth, dirs):
        self.color[row].copy_(color.reshape(self.P, 3)); self.depth[row].copy_(depth.reshape(self.P)); self.dirs[row].copy_(dirs.reshape(self.P, 3))

    def cut(self, row, color, depth, dirs, seed):
        """row <- a random subset of P distinct pixels of the frame (color [H,W,3], depth [H,W], dirs [H,W,3]): Mapper.py:329-337's
        randperm + gathers as ONE launch (us_pool_cut).  Returns a device int32[1]: 1 if the pool holds a pixel without a depth."""
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        c, d, r = L.f32(color), L.f32(depth), L.f32(dirs)
        L.check(L.lib().us_pool_cut(L.ptr(c), L.ptr(d), L.ptr(r), d.numel(), self.P, int(seed) & (2 ** 64 - 1), L.ptr(self.color[row]), L.ptr(self.depth[row]),
                                    L.ptr(self.dirs[row]), L.ptr(flag), L.stream()), "us_pool_cut")
        return flag


class ArenaWindow(MapWindow):
    """
    MapWindow over a KeyframeArena with the window's SHAPE on the device (us_arena_window_sample / us_arena_pose_step): a fixed row layout
    -- rows_a rays for the frames' shares, rows_b for the extra rays of the newest frames -- whatever the number of frames, so ONE captured
    graph serves every mapped frame of a run: bind() writes the window (arena rows, poses, shape: a few hundred bytes), replay() runs
    an iteration.  Rows beyond b * (rows_a // b) are padding: rendered, flagged invalid, dropped by the loss like pre-filtered rays.
    """

    def __init__(self, step, arena, rows_a, rows_b=0, joint_opt=True, cam_lr=1e-3, has_zero_depth=None):
        self.step, self.arena = step, arena
        dev = step.device
        self.P, self.cap = arena.P, arena.K
        self.rows_a, self.rows_b = int(rows_a), int(rows_b)
        self.R_a, self.R = self.rows_a, self.rows_a + self.rows_b
        self.first, self.joint_opt, self.cam_lr = 1, bool(joint_opt), float(cam_lr)
        self.b, self.n_per, self.extra = 1, self.rows_a, None     # host mirror of the bound window (bind() sets it)
        self.shape_dev = torch.zeros(8, dtype=torch.int32, device=dev)
        self.slots = torch.zeros(self.cap, dtype=torch.int32, device=dev)
        self._stage = [torch.zeros(8 + self.cap, dtype=torch.int32).pin_memory() for _ in range(4)]
        self._stage_ev = [None] * len(self._stage)                  # the copy out of stage k: waited for before the host writes it again
        self._stage_k = 0
        self.c2w_first = torch.eye(4, device=dev)
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.poses, self.pm, self.pv, self.g_pose = f(self.cap, 7), f(self.cap, 7), f(self.cap, 7), f(self.cap, 7)
        self.poses[:, 0] = 1.0
        self.ro, self.rd, self.gd, self.gc, self.dirs = f(self.R, 3), f(self.R, 3), f(self.R), f(self.R, 3), f(self.R, 3)
        self.idx_a = torch.zeros(self.rows_a, dtype=torch.int64, device=dev)
        self.idx_b = torch.zeros(max(self.rows_b, 1), dtype=torch.int64, device=dev)
        self.has_zero = True if has_zero_depth is None else bool(has_zero_depth)
        self._graph, self.t_rand, self.zd_draws = None, None, None
        self._comm, self._shard = None, None
        self.generation = arena.generation
        self._write_shape([0], self.rows_a, None)
        if self.R > step.max_rays:
            step._alloc(self.R)

    def _write_shape(self, slots, n_per, extra):
        b = len(slots)
        if b > self.cap:
            raise L.UniSlamHipError(f"ArenaWindow: {b} frames, capacity {self.cap}")
        xf, xn = (min(int(extra[0]), b), int(extra[1])) if extra else (0, 0)
        if b * n_per > self.rows_a or xf * xn > self.rows_b or n_per < 1:
            raise L.UniSlamHipError(f"ArenaWindow: window of {b} x {n_per} + {xf} x {xn} rays does not fit rows {self.rows_a} + {self.rows_b}")
        k = self._stage_k
        st = self._stage[k]; self._stage_k = (k + 1) % len(self._stage)
        if self._stage_ev[k] is not None:
            self._stage_ev[k].synchronize()                          # (four binds ago: normally long done)
        st[:8] = torch.tensor([b, n_per, xf, xn, 1, 0, 0, 0], dtype=torch.int32)
        st[8:8 + b] = torch.as_tensor(slots, dtype=torch.int32)
        self.shape_dev.copy_(st[:8], non_blocking=True)
        self.slots[:b].copy_(st[8:8 + b], non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        self._stage_ev[k] = ev
        self.b, self.n_per, self.extra = b, int(n_per), ((xf, xn) if xf and xn else None)

    def bind(self, slots, c2ws, n_per, extra=None):
        """the next window: arena rows of its frames (oldest first, the frame being mapped last), their poses [b,4,4], pixels per frame,
        extra = None | (n_frames, n_pixels).  Fresh pose moments (src/Mapper.py:358-364: a new optimiser per mapped frame)."""
        if self.arena.generation != self.generation:
            raise L.UniSlamHipError("ArenaWindow.bind: the arena has grown since this window (and its graph) was built; build a new one")
        self._write_shape(slots, n_per, extra)
        c2ws = L.f32(c2ws.detach().to(self.poses.device))
        self.c2w_first.copy_(c2ws[0])
        if self.b > 1:
            L.check(L.lib().us_matrix_to_cam_pose(L.ptr(c2ws[1:].contiguous()), self.b - 1, 0, L.ptr(self.poses), L.stream()), "us_matrix_to_cam_pose")
        self.pm.zero_(); self.pv.zero_()
        return self

    def draw(self, indices=None, indices_extra=None):
        if indices is None:
            torch.randint(self.P, (self.rows_a,), device=self.idx_a.device, out=self.idx_a)
        else:
            self.idx_a[:indices.numel()].copy_(indices.reshape(-1))
        if self.rows_b:
            if indices_extra is None:
                torch.randint(self.P, (self.rows_b,), device=self.idx_b.device, out=self.idx_b)
            else:
                self.idx_b[:indices_extra.numel()].copy_(indices_extra.reshape(-1))

    def rays(self):
        raise L.UniSlamHipError("ArenaWindow: rays are formed inside the iteration (us_arena_window_sample)")

    def _sample(self, tr, device_draw):
        lib, st, P, s, a = L.lib(), L.stream(), L.ptr, self.step, self.arena
        ia, ib = (None, None) if device_draw else (P(self.idx_a), P(self.idx_b))
        L.check(lib.us_arena_window_sample(P(self.c2w_first), P(self.poses), P(self.shape_dev), P(self.slots), self.rows_a, self.rows_b, P(a.depth),
                                           P(a.color), P(a.dirs), self.P, ia, ib, s.bhost, P(s.t_uni), s.n_strat, P(s.t_surf), s.n_imp,
                                           ctypes.c_float(1.2), ctypes.c_float(1.5 * s.truncation), ctypes.c_float(3 * s.truncation), tr, s._seed(0),
                                           P(s.step_dev), 1 if s.perturb else 0, P(self.ro), P(self.rd), P(self.dirs), P(self.gd), P(self.gc),
                                           P(s.valid), P(s.z), P(s.pts), st), "us_arena_window_sample")

    def _pose_desc(self):
        P, s = (lambda t: ctypes.c_void_p(t.data_ptr())), self.step
        return L.PoseStepDesc(P(self.poses), self.cap, P(s.g_o), P(s.g_d), P(self.dirs), 0, 0, 0, 0, 0, P(self.pm), P(self.pv), P(self.g_pose),
                              self.cam_lr, self.cam_lr, P(self.shape_dev), self.rows_a)

    def _pose_step(self):
        lib, st, P, s = L.lib(), L.stream(), L.ptr, self.step
        if not s._step_advanced:
            L.check(lib.us_adam_step_inc(P(s.step_dev), 0.9, 0.999, st), "us_adam_step_inc")
            s._step_advanced = True
        lr = self.cam_lr
        L.check(lib.us_arena_pose_step(P(self.poses), self.cap, P(self.shape_dev), self.rows_a, P(s.g_o), P(s.g_d), P(self.dirs), P(self.pm), P(self.pv),
                                       P(self.g_pose), lr, lr, 0.9, 0.999, 1e-8, P(s.step_dev), st), "us_arena_pose_step")

    def c2ws(self):
        opt = [cam_pose_to_matrix(self.poses[:self.b - 1])] if self.b > 1 else []
        return torch.cat([self.c2w_first[None]] + opt, dim=0).clone()
