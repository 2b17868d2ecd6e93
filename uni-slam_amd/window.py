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
synchronisation, so it can be captured into ONE hipGraph (capture() / replay(); the pixel draw stays outside the graph).
With joint_opt off the window is the same machine without the pose step (the poses then never move).
"""
import ctypes

import torch

from . import _lib as L
from .common import cam_pose_to_matrix, matrix_to_cam_pose


class MapWindow:
    def __init__(self, step, c2ws, depths, colors, dirs, n_per_frame, joint_opt=True, cam_lr=1e-3, extra=None, has_zero_depth=None):
        """
        step: MapStep (owns the model, the optimiser state and the render buffers; call step.reset_optimizer() first, as
              Mapper.optimize_mapping builds a fresh Adam per mapped frame, src/Mapper.py:358-364);
        c2ws [b,4,4]: the window's poses, the OLDEST first -- it stays fixed (src/Mapper.py:374); depths [b,P], colors [b,P,3],
        dirs [b,P,3]: the frames' pixel pools (camera-frame directions); n_per_frame = mapping_pixels // b (src/Mapper.py:315);
        extra: None | (n_frames, n_pixels): n_pixels more rays from each of the newest n_frames frames (src/Mapper.py:385-393: 10 x 200
               once the keyframe list has more than 20 entries and the tracker is not tracking back);
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
        self.joint_opt = bool(joint_opt) and b > 1
        self.cam_lr = float(cam_lr)
        self.pool_d, self.pool_c, self.pool_r = L.f32(depths.to(dev)), L.f32(colors.to(dev)), L.f32(dirs.to(dev))
        c2ws = L.f32(c2ws.detach().to(dev))
        self.c2w_first = c2ws[0].clone()
        n_p = max(b - 1, 1)
        self.poses = matrix_to_cam_pose(c2ws[1:]).contiguous() if b > 1 else torch.zeros((1, 7), device=dev)   # src/Mapper.py:360
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
        poses = P(self.poses) if b > 1 else None
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

    def _launches(self, t_rand=None, zero_depth_draws=None, device_draw=False):
        """everything after the draw: rays from the current poses + samples (ONE launch: us_window_sample; with device_draw it also
        draws the pixels, like the jitter from a counter-based generator salted with the device-side step count), render + loss +
        backward, pose step, Adam"""
        lib, st, P, s = L.lib(), L.stream(), L.ptr, self.step
        b = self.b
        if t_rand is None:
            t_rand = self.t_rand                                 # the static jitter tensor of capture(t_rand=True), if any
        if zero_depth_draws is None:
            zero_depth_draws = self.zd_draws                     # ... and the static draws of the zero-depth branch
        poses = P(self.poses) if b > 1 else None
        nf, ne = self.extra if self.extra else (0, 0)
        if self.R > s.max_rays:
            s._alloc(self.R)
        s.rng_calls += 1
        tr = P(L.f32(t_rand)) if (s.perturb and t_rand is not None) else None
        ia, ib = (None, None) if device_draw else (P(self.idx_a), P(self.idx_b) if self.extra else None)
        L.check(lib.us_window_sample(P(self.c2w_first), poses, b, self.n_per, nf, ne, P(self.pool_d), P(self.pool_c), P(self.pool_r), self.P, ia, ib,
                                     s.bhost, P(s.t_uni), s.n_strat, P(s.t_surf), s.n_imp, ctypes.c_float(1.2), ctypes.c_float(1.5 * s.truncation),
                                     ctypes.c_float(3 * s.truncation), tr, s._seed(0), P(s.step_dev), 1 if s.perturb else 0, P(self.ro),
                                     P(self.rd), P(self.dirs), P(self.gd), P(self.gc), P(s.valid), P(s.z), P(s.pts), st), "us_window_sample")
        if not self.joint_opt:
            return s.iterate(self.ro, self.rd, self.gd, self.gc, t_rand=t_rand, has_zero_depth=self.has_zero, presampled=True,
                             zero_depth_draws=zero_depth_draws)
        if s.group is not None:
            raise L.UniSlamHipError("MapWindow: joint pose optimisation runs in a single process (the poses are not all-reduced)")
        s.store_dydx = True                                      # the encoder leaves dy/dx for the pose gradient (no second gather pass)
        try:
            s.forward(self.ro, self.rd, self.gd, self.gc, t_rand, self.has_zero, zero_depth_draws, presampled=True)
        finally:
            s.store_dydx = False
        loss = s.backward(ray_grads=True, fold=True)              # (adam_step() below sums the decoder-gradient partials itself)
        g_o, g_d = s.g_o, s.g_d
        if not s._step_advanced:                                 # the poses are one more group of the SAME optimiser: one step count
            L.check(lib.us_adam_step_inc(P(s.step_dev), 0.9, 0.999, st), "us_adam_step_inc")
            s._step_advanced = True
        lr = self.cam_lr                                         # the pose group is appended with its plain lr (src/Mapper.py:362): no lr_factor
        # pose j = window frame j + 1: rows [(j+1) n_per, (j+2) n_per) of the first block, and rows of the extra block for the newest nf frames
        L.check(lib.us_pose_window_step(P(self.poses), b - 1, P(g_o), P(g_d), P(self.dirs), self.n_per, self.n_per, max(b - nf - 1, 0),
                                        self.R_a + (ne if (nf == b and nf > 0) else 0), ne if nf else 0, P(self.pm), P(self.pv), P(self.g_pose),
                                        lr, lr, 0.9, 0.999, 1e-8, P(s.step_dev), 0, st), "us_pose_window_step")
        s.adam_step()
        return loss

    def iterate(self, indices=None, indices_extra=None, t_rand=None, zero_depth_draws=None):
        """one eager iteration; returns the loss tensor [1] (device).  indices None: the pixels are drawn inside us_window_sample"""
        if indices is not None:
            self.draw(indices, indices_extra)
        return self._launches(t_rand, zero_depth_draws, device_draw=indices is None)

    # ------------------------------------------------------------------------------------------ hipGraph
    def capture(self, t_rand=False, device_draw=True):
        """capture _launches() (the zero-depth branch included: its row count stays on the device) into a hipGraph: replay() is ONE graph
        launch, pixel draw included (device_draw; False: replay(indices) / torch.randint fill the static index tensors first).  The
        jitter comes from the in-kernel generator (varied per replay by the device-side step count) unless t_rand=True: then
        self.t_rand [R,S] (and, for a window with depth holes, self.zd_draws) are static inputs to fill.  The model, the optimiser state and the poses are left as they were."""
        from .graph import CapturedIteration
        s = self.step
        if s.group is not None:
            raise L.UniSlamHipError("MapWindow.capture: single-process only")
        if s._step_advanced:
            raise L.UniSlamHipError("MapWindow.capture: a forward pass of the MapStep is pending; finish its optimiser step first")
        s._join_side_streams()
        was, s.probe = s.probe, None
        self.t_rand = torch.zeros((self.R, s.S), dtype=torch.float32, device=s.device) if t_rand else None
        # (static draws of the zero-depth branch, indexed by the compacted row: jitter of the coarse pass [R, n_strat], inverse-transform draws [R, n_imp])
        self.zd_draws = (torch.zeros((self.R, s.n_strat), device=s.device), torch.zeros((self.R, s.n_imp), device=s.device)) if (t_rand and self.has_zero) else None
        keep = (s.flat.clone(), s.m.clone(), s.v.clone(), s.step_dev.clone(), s.opt_step, dict(s.lr), s.rng_calls, self.poses.clone(),
                self.pm.clone(), self.pv.clone(), self.cam_lr)
        s.lr = {k: 0.0 for k in s.lr}
        self.cam_lr = 0.0
        self.draw()
        self._device_draw = bool(device_draw)
        run = lambda: self._launches(device_draw=self._device_draw)
        try:
            side = torch.cuda.Stream(device=s.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    run()
            torch.cuda.current_stream().wait_stream(side)
        finally:
            s.lr, self.cam_lr = keep[5], keep[10]
        s.flat.copy_(keep[0]); s.m.copy_(keep[1]); s.v.copy_(keep[2]); s.step_dev.copy_(keep[3])
        self.poses.copy_(keep[7]); self.pm.copy_(keep[8]); self.pv.copy_(keep[9])
        s.opt_step, s.rng_calls = keep[4], keep[6]
        s._dec_grad_clean = False
        self._graph = CapturedIteration(run, warmup=0)
        s.opt_step = keep[4]
        s.probe = was

    def replay(self, indices=None, indices_extra=None):
        if self._graph is None:
            raise L.UniSlamHipError("MapWindow.replay: call capture() first")
        if self._device_draw:
            if indices is not None:
                raise L.UniSlamHipError("MapWindow.replay: this graph draws its pixels itself; capture(device_draw=False) to pass indices")
        else:
            self.draw(indices, indices_extra)
        self.step.opt_step += 1
        return self._graph.replay()

    # ------------------------------------------------------------------------------------------ results
    def c2ws(self):
        """the window's poses now: [b,4,4] (frame 0 as given; the others from the optimised quaternion / translation, src/Mapper.py:449)"""
        if self.b == 1:
            return self.c2w_first[None].clone()
        return torch.cat([self.c2w_first[None], cam_pose_to_matrix(self.poses)], dim=0)
