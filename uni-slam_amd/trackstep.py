"""
TrackStep -- the body of the reference's tracking hot loop (src/Tracker.py:149-244) as a straight-line sequence of HIP launches.
"""
import ctypes

import torch

from . import _lib as L
from .common import bound_host
from .decoders import Decoders
from .hashgrid import HashGridEncoding


class TrackStep:
    """
    The body of the reference's tracking hot loop (src/Tracker.py:149-244, one call of optimize_tracking) with the
    render + loss + backward part as a straight-line sequence of HIP launches: decoders and tables are frozen, the only
    gradient wanted is the one of the camera pose, which arrives as dL/d(rays_o), dL/d(rays_d) and is pushed through the
    tiny quaternion -> rotation -> ray graph by torch autograd.  No boolean compaction (validity flags instead), no table
    or decoder gradients (the reference computes and discards them, Tracker.py:110-111), one host-free median.
    """

    def __init__(self, hash_grid_sdf, hash_grid_color, decoders, bound, n_stratified, n_importance, truncation, weights,
                 mask_mode="original", perturb=True, max_rays=2048):
        assert isinstance(hash_grid_sdf, HashGridEncoding) and isinstance(hash_grid_color, HashGridEncoding)
        self.es, self.ec, self.dec = hash_grid_sdf, hash_grid_color, decoders
        dev = hash_grid_sdf.params.device
        if dev.type != "cuda":
            raise L.UniSlamHipError("TrackStep needs the model on the GPU")
        self.device = dev
        self.S, self.n_strat, self.n_imp = n_stratified + n_importance, n_stratified, n_importance
        self.truncation = float(truncation)
        self.w5 = L.host_floats([weights["fs"], weights["center"], weights["tail"], weights["color"], weights["depth"]])
        self.mode = {"original": 2, "no_mask": 3}[mask_mode]
        self.perturb = perturb
        self.bound = bound.to(dev)
        self.bhost = bound_host(bound)
        self.t_uni = torch.linspace(0., 1., steps=n_stratified).to(dev)
        self.t_surf = torch.linspace(0., 1., steps=n_importance).to(dev)
        self.desc_s, self.desc_c = decoders.mlp_descs()
        self._joint = None
        self._alloc(max_rays)

    def _alloc(self, R):
        # generation: bumped whenever a buffer a captured graph may hold changes its address (holders compare it before a replay)
        self.generation = getattr(self, "generation", 0) + 1
        dev, S = self.device, self.S
        N = R * S
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.max_rays = R
        self.dydx_s = self.dydx_c = self.dpts_s = self.dpts_c = None
        self.z, self.pts, self.d_pts = f(R, S), f(R, S, 3), f(R, S, 3)
        self.feat_s, self.feat_c, self.d_feat_s, self.d_feat_c = f(N, 32), f(N, 32), f(N, 32), f(N, 32)
        self.raw, self.d_raw = f(R, S, 4), f(R, S, 4)
        self.term, self.unc, self.depth, self.dunc, self.rgb = f(R), f(R), f(R), f(R), f(R, 3)
        self.g_sdf, self.g_depth, self.g_rgb = f(R, S), f(R), f(R, 3)
        self.g_o, self.g_d = f(R, 3), f(R, 3)
        self.partials = f(int(L.lib().us_loss_partials_size(R)))
        self.stats, self.loss, self.median, self.err = f(10), f(1), f(1), f(R)
        self.valid = torch.empty(R, dtype=torch.uint8, device=dev)

    def _split_flags(self, joint):
        """(encoder flag, decoder flag): with two split-bf16 decoders the joint encoder writes the features as the decoders' hi / lo bf16
        operand pairs (same bytes and values; the decoders skip the split of their inputs).  feat_split = False turns it off."""
        on = bool(joint and getattr(self, "feat_split", L.FEAT_SPLIT_DEFAULT) and self.desc_s.precision == 1 and self.desc_c.precision == 1
                  and self.desc_s.n_in == 32 and self.desc_c.n_in == 32)
        return (L.US_GRID_FEAT_SPLIT_BF16, L.US_MLP_IN_SPLIT_BF16) if on else (0, 0)

    def _decoder_params(self):
        dec = self.dec
        if dec.tcnn_network:
            return L.f32(dec.sdf_decoder.params.detach()), L.f32(dec.color_decoder.params.detach())
        return (Decoders.pack_linear_params(dec.linears, dec.output_linear).detach(),
                Decoders.pack_linear_params(dec.c_linears, dec.c_output_linear).detach())

    def refresh_parameters(self):
        """call after the mapper changed the decoders (Tracker.update_params_from_mapping, Tracker.py:246-269)"""
        # into buffers that keep their addresses (a captured iteration holds them): packed copies are refreshed in place, views of
        # live parameters (tcnn-layout decoders, a learnable beta) are already the parameters themselves
        def keep(name, new):
            old = getattr(self, name, None)
            if old is not None and old.shape == new.shape and old.data_ptr() != new.data_ptr():
                old.copy_(new)
            else:
                if old is not None and old.data_ptr() != new.data_ptr():
                    self.generation = getattr(self, "generation", 0) + 1        # a captured graph holds the old address
                setattr(self, name, new)
        ps, pc = self._decoder_params()
        keep("_ps", ps); keep("_pc", pc)
        b = self.dec.beta
        keep("_beta", L.f32(b.detach()).reshape(1) if torch.is_tensor(b) else torch.tensor([float(b)], device=self.device))

    def forward_backward(self, rays_o, rays_d, gt_depth, gt_color, t_rand=None):
        """render the rays, evaluate the tracking loss, return (loss[1], g_rays_o[R,3], g_rays_d[R,3], pixel_unc[R], valid[R])"""
        lib, st, P = L.lib(), L.stream(), L.ptr
        if not hasattr(self, "_ps"):
            self.refresh_parameters()
        o, d, gd, gc = L.f32(rays_o.detach()), L.f32(rays_d.detach()), L.f32(gt_depth.detach()), L.f32(gt_color.detach())
        R, S = o.shape[0], self.S
        if R > self.max_rays:
            self._alloc(R)
        N = R * S
        off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
        ds, dc = ctypes.byref(self.es.desc), ctypes.byref(self.ec.desc)
        ms, mc = ctypes.byref(self.desc_s), ctypes.byref(self.desc_c)
        ts, tc = L.f32(self.es.params.detach()), L.f32(self.ec.params.detach())
        # Tracker.py:177-184 (inside the box AND a depth measurement, kept as flags) + Renderer.py:81-101,132-137 in one launch;
        # jitter from t_rand or from the in-kernel generator (the device-side step count varies it between graph replays)
        tr = P(L.f32(t_rand)) if (self.perturb and t_rand is not None) else None
        self.rng_calls = getattr(self, "rng_calls", 0) + 1
        seed = (int(torch.initial_seed()) + 0x9E3779B97F4A7C15 * self.rng_calls) & (2 ** 64 - 1)
        L.check(lib.us_sample_points(P(o), P(d), P(gd), self.bhost, R, P(self.t_uni), self.n_strat, P(self.t_surf), self.n_imp,
                                     ctypes.c_float(1.2), ctypes.c_float(1.5 * self.truncation), ctypes.c_float(3 * self.truncation), tr, seed,
                                     P(self.draw_ctr) if hasattr(self, "draw_ctr") else None, 1 if self.perturb else 0, 1, P(self.valid),
                                     P(self.z), P(self.pts), st), "us_sample_points")
        # level-major feature planes (flags 3 = clamp + level-major); no dy_dx is stored: the pose gradient re-gathers.  Both tables in
        # one launch where the pair of grids qualifies (positions and cells computed once, one launch less in a latency-bound chain)
        if self._joint is None:
            self._joint = bool(lib.us_hashgrid_joint_supported(ds, dc, 1))    # the count-free encoder needs the shared geometry only
        gs_, ms_ = self._split_flags(self._joint)
        if self._joint:
            # ... and d(features)/d(position) of both, which the pose gradient contracts at the end (no second gather pass over the tables)
            if self.dydx_s is None:
                self.dydx_s = torch.empty(self.es.desc.n_levels * self.max_rays * S * 6, dtype=torch.float16, device=self.device)
                self.dydx_c = torch.empty_like(self.dydx_s)
            L.check(lib.us_hashgrid_fwd_joint_dydx(ds, dc, P(ts), P(tc), P(self.pts), N, P(self.feat_s), P(self.feat_c), P(self.dydx_s), P(self.dydx_c),
                                                   3 | gs_, None, 0, st), "us_hashgrid_fwd_joint_dydx")
        else:
            L.check(lib.us_hashgrid_fwd(ds, P(ts), P(self.pts), N, P(self.feat_s), None, 3, st), "us_hashgrid_fwd")
            L.check(lib.us_hashgrid_fwd(dc, P(tc), P(self.pts), N, P(self.feat_c), None, 3, st), "us_hashgrid_fwd")
        pair = bool(lib.us_mlp_pair_supported(ms, mc))             # both decoders in one launch each way
        if pair:
            L.check(lib.us_mlp_fwd_pair(ms, mc, P(self._ps), P(self._pc), P(self.feat_s), P(self.feat_c), N, off(self.raw, 3), 4, P(self.raw), 4, 1 | ms_, st),
                    "us_mlp_fwd_pair")
        else:
            L.check(lib.us_mlp_fwd(ms, P(self._ps), P(self.feat_s), N, off(self.raw, 3), 4, 1 | ms_, st), "us_mlp_fwd")
            L.check(lib.us_mlp_fwd(mc, P(self._pc), P(self.feat_c), N, P(self.raw), 4, 1 | ms_, st), "us_mlp_fwd")
        L.check(lib.us_composite_fwd(P(self.raw), P(self.z), P(self._beta), R, S, P(self.term), P(self.unc), P(self.depth),
                                     P(self.rgb), P(self.dunc), None, st), "us_composite_fwd")
        med = None
        if self.mode == 2:
            # Tracker.py:214-215: median of |gt - depth| over the rays that passed the pre-filter (lower median, like
            # torch.median), without compaction: rejected rays sort to the end as +inf
            if R <= 8192:
                L.check(lib.us_masked_median(P(gd), P(self.depth), P(self.valid), R, P(self.median), st), "us_masked_median")
            else:
                valid = self.valid[:R].bool()
                err = torch.where(valid, (gd - self.depth[:R]).abs(), torch.full_like(gd, float("inf")))
                k = torch.clamp((valid.sum() - 1) // 2, min=0)
                self.median.copy_(torch.sort(err)[0].gather(0, k.reshape(1)))
            med = P(self.median)
        L.check(lib.us_loss_stats(self.mode, off(self.raw, 3), 4, P(self.valid), P(self.z), P(gd), P(gc), P(self.depth), P(self.rgb),
                                  P(self.unc), med, R, S, self.truncation, P(self.partials), P(self.stats), st), "us_loss_stats")
        L.check(lib.us_loss_grad(self.mode, off(self.raw, 3), 4, P(self.valid), P(self.z), P(gd), P(gc), P(self.depth), P(self.rgb),
                                 P(self.unc), med, R, S, self.truncation, self.w5, P(self.stats), P(self.g_sdf), P(self.g_depth),
                                 P(self.g_rgb), P(self.loss), st), "us_loss_grad")
        L.check(lib.us_composite_bwd(P(self.raw), P(self.z), P(self._beta), R, S, None, None, P(self.g_depth), P(self.g_rgb), None,
                                     P(self.g_sdf), P(self.d_raw), None, None, st), "us_composite_bwd")
        if pair:
            L.check(lib.us_mlp_bwd_pair(ms, mc, P(self._ps), P(self._pc), P(self.feat_s), P(self.feat_c), off(self.raw, 3), 4, P(self.raw), 4,
                                        off(self.d_raw, 3), 4, P(self.d_raw), 4, N, P(self.d_feat_s), P(self.d_feat_c), None, None, 1 | ms_, None, None, 0, st),
                    "us_mlp_bwd_pair")
        else:
            L.check(lib.us_mlp_bwd(ms, P(self._ps), P(self.feat_s), off(self.raw, 3), 4, off(self.d_raw, 3), 4, N, P(self.d_feat_s), None, 1 | ms_,
                                   None, 0, st), "us_mlp_bwd")
            L.check(lib.us_mlp_bwd(mc, P(self._pc), P(self.feat_c), P(self.raw), 4, P(self.d_raw), 4, N, P(self.d_feat_c), None, 1 | ms_,
                                   None, 0, st), "us_mlp_bwd")
        if self._joint and S <= 128:
            L.check(lib.us_hashgrid_dydx_rays(self.es.desc.n_levels, P(self.d_feat_s), P(self.d_feat_c), P(self.dydx_s), P(self.dydx_c), R, S, P(self.z),
                                              self.bhost, P(self.g_o), P(self.g_d), None, st), "us_hashgrid_dydx_rays")
        elif lib.us_hashgrid_bwd_input_rays_supported(ds, dc, S):
            # both grids' input gradient and its reduction to the rays in one launch (was: two gathers + us_ray_points_bwd)
            L.check(lib.us_hashgrid_bwd_input_rays(ds, dc, P(ts), P(tc), P(self.pts), P(self.d_feat_s), P(self.d_feat_c), R, S, P(self.z), self.bhost,
                                                   P(self.g_o), P(self.g_d), None, 3, st), "us_hashgrid_bwd_input_rays")
        else:
            L.check(lib.us_hashgrid_bwd_input_gather(ds, P(ts), P(self.pts), P(self.d_feat_s), N, P(self.d_pts), 3, st),
                    "us_hashgrid_bwd_input_gather")
            L.check(lib.us_hashgrid_bwd_input_gather(dc, P(tc), P(self.pts), P(self.d_feat_c), N, P(self.d_pts), 3 | L.US_GRID_ACCUMULATE, st),
                    "us_hashgrid_bwd_input_gather")
            L.check(lib.us_ray_points_bwd(P(self.d_pts), P(self.z), self.bhost, R, S, P(self.g_o), P(self.g_d), st), "us_ray_points_bwd")
        return self.loss, self.g_o[:R], self.g_d[:R], self.unc[:R], self.valid[:R]

    def iterate(self, cam_pose, gt_color, gt_depth, batch_size, optimizer, H, W, fx, fy, cx, cy, ignore_edge_H, ignore_edge_W,
                t_rand=None, indices=None):
        """
        Tracker.optimize_tracking (Tracker.py:149-244): cam_pose [1,7] (quaternion, translation) with requires_grad,
        gt_color [1,H,W,3], gt_depth [1,H,W].  Returns (loss tensor[1], pixel_unc of the rays that passed the pre-filter
        mask applied as in the reference is left to the caller: pixel_unc[valid]).
        """
        from .common import cam_pose_to_matrix, get_rays_from_uv
        dev = self.device
        c2w = cam_pose_to_matrix(cam_pose)
        H0, H1, W0, W1 = ignore_edge_H, H - ignore_edge_H, ignore_edge_W, W - ignore_edge_W
        n_pix = (H1 - H0) * (W1 - W0)
        if indices is None:
            indices = torch.randint(n_pix, (batch_size,), device=dev)                     # common.py:116
        # pixel (i, j) of flat crop index: i = W0 + idx % (W1-W0), j = H0 + idx // (W1-W0)   (common.py:144-148)
        wi = W1 - W0
        i = (W0 + indices % wi).float()[None]
        j = (H0 + torch.div(indices, wi, rounding_mode="floor")).float()[None]
        gd = gt_depth[0, H0:H1, W0:W1].reshape(-1)[indices]
        gc = gt_color[0, H0:H1, W0:W1].reshape(-1, 3)[indices]
        rays_o, rays_d = get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy, dev)
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        loss, g_o, g_d, unc, valid = self.forward_backward(rays_o, rays_d, gd, gc, t_rand)
        optimizer.zero_grad()
        torch.autograd.backward([rays_o, rays_d], [g_o, g_d])
        optimizer.step()
        return loss, unc, valid

    # ------------------------------------------------------------------------------------------ fully fused tracking
    def begin_frame(self, pose7, gt_color, gt_depth, lr_T, lr_R, H, W, fx, fy, cx, cy, ignore_edge_H, ignore_edge_W,
                    betas=(0.5, 0.999), refresh=True):
        """
        Per-frame set-up of the fused tracking loop (Tracker.py:315-329): pose7 = (quaternion[4], translation[3]) initial
        guess, gt_color [H,W,3], gt_depth [H,W]; a fresh Adam state for the two parameter groups (lr_R for the quaternion,
        lr_T for the translation).  The buffers are static, so iterate_fused() can be captured into a hipGraph.  refresh: re-read the
        decoders' parameters (Tracker.update_params_from_mapping, Tracker.py:246-269).
        """
        dev = self.device
        if not hasattr(self, "pose"):
            f = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
            self.pose, self.g_pose, self.pm, self.pv, self.pstep = f(7), f(7), f(7), f(7), f(1)
            self.img_d, self.img_c = torch.empty((H, W), device=dev), torch.empty((H, W, 3), device=dev)
        if not hasattr(self, "min_loss"):
            self.min_loss, self.best_pose, self.mean_unc, self.draw_ctr = (torch.zeros(k, dtype=torch.float32, device=dev) for k in (1, 7, 1, 1))
        self.pose.copy_(pose7.detach().reshape(7))
        self.min_loss.fill_(float("inf")); self.best_pose.copy_(self.pose)     # the loop's candidate (Tracker.py:331,346-348)
        self.img_d.copy_(gt_depth.reshape(H, W)); self.img_c.copy_(gt_color.reshape(H, W, 3))
        self.pm.zero_(); self.pv.zero_(); self.pstep.zero_()
        self.lr_T, self.lr_R, self.betas = float(lr_T), float(lr_R), betas
        self.frame = (H, W, ignore_edge_H, ignore_edge_W)
        self.intr = L.host_floats([fx, fy, cx, cy])
        if refresh or not hasattr(self, "_ps"):                  # refresh=False: the caller knows the decoders have not changed since
            self.refresh_parameters()

    def iterate_fused(self, batch_size, t_rand=None, indices=None):
        """
        One Tracker.optimize_tracking call (Tracker.py:149-244) with everything on the device.  Returns (loss[1], pixel_unc[R],
        valid[R]); the updated pose is self.pose.  The loop's minimum-loss bookkeeping (Tracker.py:346-348) rides on the pose step's
        launch (us_pose_track_step): self.min_loss[1] / self.best_pose[7] (reset by begin_frame) take this iteration's loss and the pose
        it was rendered at where the loss is the lowest so far.  The in-kernel pixel draw is keyed to self.draw_ctr, a device counter
        that the same launch advances and that is NOT reset per frame (the optimiser's own count is).  Where the model qualifies (two F = 2 grids of one geometry, a decoder pair, the
        'original' mask, <= 8192 rays of <= 128 samples) the iteration is NINE launches:
            us_track_sample              pixel draw (indices None) + pose -> rays + pre-filter + z + points
            us_hashgrid_fwd_joint_dydx   both encoders + d(features)/d(position)
            us_mlp_fwd_pair              both decoders
            us_track_loss_fwd            compositing + per-ray loss partials | median gate + statistics (one workgroup)
            us_track_loss_bwd            loss gradients + compositing backward
            us_mlp_bwd_pair_dydx         both decoders' input gradients, contracted in registers with dy/dx: dL/d(points) per grid
            us_ray_points_bwd2           the two shares added and reduced to dL/d(rays_o), dL/d(rays_d)
            us_pose_track_step           pose gradient + Adam on the 7 numbers (step count included) + minimum-loss candidate
        otherwise the general chain (us_pose_rays + forward_backward + the pose step).
        """
        lib, st, P = L.lib(), L.stream(), L.ptr
        H, W, eh, ew = self.frame
        H0, H1, W0, W1 = eh, H - eh, ew, W - ew
        n = int(batch_size)
        if n > self.max_rays:
            self._alloc(n)
        S, N = self.S, n * self.S
        ds, dc = ctypes.byref(self.es.desc), ctypes.byref(self.ec.desc)
        ms, mc = ctypes.byref(self.desc_s), ctypes.byref(self.desc_c)
        if self._joint is None:
            self._joint = bool(lib.us_hashgrid_joint_supported(ds, dc, 1))    # the count-free encoder needs the shared geometry only
        fast = (getattr(self, "fast_path", True) and self._joint and self.mode == 2 and n <= 8192 and S <= 128 and bool(lib.us_mlp_pair_supported(ms, mc)))
        b1, b2 = self.betas
        if not fast:
            if indices is None:
                indices = torch.randint((H1 - H0) * (W1 - W0), (n,), device=self.device)              # common.py:116
            if not hasattr(self, "t_ro") or self.t_ro.shape[0] != n:
                f = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
                self.t_ro, self.t_rd = f(n, 3), f(n, 3)
            self._track_inputs(n)
            L.check(lib.us_pose_rays(P(self.pose), P(indices.contiguous()), n, self.intr, W0, H0, W1 - W0, P(self.img_d), P(self.img_c), W,
                                     P(self.t_ro), P(self.t_rd), P(self.t_dirs), P(self.t_gd), P(self.t_gc), st), "us_pose_rays")
            loss, g_o, g_d, unc, valid = self.forward_backward(self.t_ro, self.t_rd, self.t_gd, self.t_gc, t_rand)
            L.check(lib.us_pose_track_step(P(self.pose), P(self.g_o), P(self.g_d), P(self.t_dirs), n, P(self.pm), P(self.pv), P(self.g_pose),
                                           self.lr_R, self.lr_T, b1, b2, 1e-8, P(self.pstep), P(loss), P(self.min_loss), P(self.best_pose),
                                           P(self.draw_ctr), st), "us_pose_track_step")
            return loss, unc, valid
        self._track_inputs(n)
        off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
        ts, tc = L.f32(self.es.params.detach()), L.f32(self.ec.params.detach())
        tr = P(L.f32(t_rand)) if (self.perturb and t_rand is not None) else None
        self.rng_calls = getattr(self, "rng_calls", 0) + 1
        seed = (int(torch.initial_seed()) + 0x9E3779B97F4A7C15 * self.rng_calls) & (2 ** 64 - 1)
        pix = P(indices.contiguous()) if indices is not None else None
        T = self._timed
        T("us_track_sample", lambda: lib.us_track_sample(
            P(self.pose), pix, n, self.intr, W0, H0, W1 - W0, H1 - H0, P(self.img_d), P(self.img_c), W, self.bhost, P(self.t_uni), self.n_strat,
            P(self.t_surf), self.n_imp, ctypes.c_float(1.2), ctypes.c_float(1.5 * self.truncation), ctypes.c_float(3 * self.truncation), tr, seed,
            P(self.draw_ctr), 1 if self.perturb else 0, None, None, P(self.t_dirs), P(self.t_gd), P(self.t_gc), P(self.valid), P(self.z), P(self.pts), st))
        if self.dydx_s is None:
            self.dydx_s = torch.empty(self.es.desc.n_levels * self.max_rays * S * 6, dtype=torch.float16, device=self.device)
            self.dydx_c = torch.empty_like(self.dydx_s)
        gs_, ms_ = self._split_flags(True)
        T("us_hashgrid_fwd_joint_dydx", lambda: lib.us_hashgrid_fwd_joint_dydx(
            ds, dc, P(ts), P(tc), P(self.pts), N, P(self.feat_s), P(self.feat_c), P(self.dydx_s), P(self.dydx_c), 3 | gs_, None, 0, st))
        T("us_mlp_fwd_pair", lambda: lib.us_mlp_fwd_pair(
            ms, mc, P(self._ps), P(self._pc), P(self.feat_s), P(self.feat_c), N, off(self.raw, 3), 4, P(self.raw), 4, 1 | ms_, st))
        T("us_track_loss_fwd", lambda: lib.us_track_loss_fwd(
            P(self.raw), P(self.z), P(self._beta), n, S, P(self.valid), P(self.t_gd), P(self.t_gc), self.truncation, P(self.term), P(self.unc),
            P(self.depth), P(self.rgb), P(self.dunc), P(self.partials), P(self.err), P(self.median), P(self.stats), st))
        T("us_track_loss_bwd", lambda: lib.us_track_loss_bwd(
            P(self.raw), P(self.z), P(self._beta), n, S, P(self.valid), P(self.t_gd), P(self.t_gc), P(self.depth), P(self.rgb), P(self.unc),
            P(self.median), self.truncation, self.w5, P(self.stats), P(self.d_raw), P(self.loss), st))
        # decoder pair backward: the input gradients stay in registers and are contracted there with dy/dx (each grid's share of
        # dL/d(point)); nothing else reads dL/d(features) in tracking, so it is not written
        if getattr(self, "dpts_s", None) is None or self.dpts_s.numel() < self.max_rays * S * 3:
            self.dpts_s = torch.empty(self.max_rays * S * 3, dtype=torch.float32, device=self.device)
            self.dpts_c = torch.empty_like(self.dpts_s)
        T("us_mlp_bwd_pair_dydx", lambda: lib.us_mlp_bwd_pair_dydx(
            ms, mc, P(self._ps), P(self._pc), P(self.feat_s), P(self.feat_c), off(self.raw, 3), 4, P(self.raw), 4, off(self.d_raw, 3), 4,
            P(self.d_raw), 4, N, None, None, None, None, 1 | ms_, None, None, 0, P(self.dydx_s), P(self.dydx_c), P(self.dpts_s), P(self.dpts_c), st))
        T("us_ray_points_bwd2", lambda: lib.us_ray_points_bwd2(P(self.dpts_s), P(self.dpts_c), P(self.z), self.bhost, n, S, P(self.g_o), P(self.g_d), st))
        T("us_pose_track_step", lambda: lib.us_pose_track_step(
            P(self.pose), P(self.g_o), P(self.g_d), P(self.t_dirs), n, P(self.pm), P(self.pv), P(self.g_pose), self.lr_R, self.lr_T,
            b1, b2, 1e-8, P(self.pstep), P(self.loss), P(self.min_loss), P(self.best_pose), P(self.draw_ctr), st))
        return self.loss, self.unc[:n], self.valid[:n]

    def probe_valid(self, batch_size, indices):
        """us_track_sample ALONE for the given pixels at the current pose: the pre-filter flags [n] (uint8, device) the next iterate_fused()
        call will compute for them.  For callers that replay a recorded random stream (slam.TorchDraws): the reference draws its jitter for
        the rays that PASSED the pre-filter only (src/Tracker.py:177-194 compacts, then src/utils/Renderer.py:55 draws [R', S]), so the
        number of rows to draw is needed before the iteration.  Nothing but this object's per-iteration scratch is written."""
        lib, st, P = L.lib(), L.stream(), L.ptr
        H, W, eh, ew = self.frame
        n = int(batch_size)
        if n > self.max_rays:
            self._alloc(n)
        self._track_inputs(n)
        L.check(lib.us_track_sample(
            P(self.pose), P(indices.contiguous()), n, self.intr, ew, eh, W - 2 * ew, H - 2 * eh, P(self.img_d), P(self.img_c), W, self.bhost, P(self.t_uni),
            self.n_strat, P(self.t_surf), self.n_imp, ctypes.c_float(1.2), ctypes.c_float(1.5 * self.truncation), ctypes.c_float(3 * self.truncation),
            None, 0, P(self.draw_ctr), 0, None, None, P(self.t_dirs), P(self.t_gd), P(self.t_gc), P(self.valid), P(self.z), P(self.pts), st),
            "us_track_sample")
        return self.valid[:n]

    def mean_uncertainty(self, unc, valid):
        """mean pixel uncertainty of the rays that passed the pre-filter (Tracker.py:353, `rendered_weights.detach().mean()`): one launch,
        the result stays on the device (self.mean_unc[1])"""
        L.check(L.lib().us_masked_mean(L.ptr(L.f32(unc)), L.ptr(valid), unc.shape[0], L.ptr(self.mean_unc), L.stream()), "us_masked_mean")
        return self.mean_unc

    def _timed(self, name, rc_fn):
        """run one C-ABI launch; with self.probe (a dict) set, bracket it with HIP events on the launch stream (bench.py)"""
        probe = getattr(self, "probe", None)
        if probe is None:
            L.check(rc_fn(), name)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(rc_fn(), name)
        e1.record()
        probe.setdefault(name, []).append((e0, e1))

    def _track_inputs(self, n):
        if not hasattr(self, "t_dirs") or self.t_dirs.shape[0] != n:
            f = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
            self.t_dirs, self.t_gd, self.t_gc = f(n, 3), f(n), f(n, 3)
