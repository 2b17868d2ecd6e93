"""
Thin tracking / mapping drivers around the hot path: the per-frame logic of the reference's
Tracker.run (src/Tracker.py:271-370) and Mapper.run / optimize_mapping / keyframe_selection_LC (src/Mapper.py:177-545) with the
inner loops served by TrackStep / MapStep.  One process, the two roles alternate (the reference runs them as two processes
that wait for each other at every `every_frame`-th frame, so the order of operations is the same).

Kept from the reference: constant-speed pose initialisation, per-frame fresh pose Adam (betas 0.5/0.999), keeping the
minimum-loss pose, the uncertainty-triggered "activated mapping" (double iterations + an extra mapping pass), keyframe pools of
10 % of the pixels, overlap-based keyframe selection incl. the loop-closure window, pixel budget per window frame, first-frame
lr factor / iterations, joint optimisation of the window's camera poses (oldest fixed).
Left out: visualiser, logger/checkpoints, mesher (SURVEY.md 8 "out of scope").
"""
import torch

from .common import cam_pose_to_matrix, get_samples, matrix_to_cam_pose, predict_cam_pose
from .mapstep import MapStep
from .trackstep import TrackStep
from .window import ArenaWindow, KeyframeArena, MapWindow

DEFAULTS = {   # configs/UNISLAM.yaml + configs/Replica/replica.yaml
    "tracking": dict(ignore_edge_W=75, ignore_edge_H=75, const_speed_assumption=True, lr_T=0.002, lr_R=0.001, pixels=2000, iters=8,
                     activated_mapping_mode=True, uncertainty_ts=0.001,
                     w=dict(fs=10, center=200, tail=50, depth=1, color=5)),
    "mapping": dict(every_frame=4, keyframe_every=4, joint_opt=True, joint_opt_cam_lr=0.001, mapping_window_size=20,
                    lr_first_factor=5, lr_factor=1, pixels=4000, iters_first=10, iters=15, LC=True, LC_ts=0.95,
                    lr=dict(decoders=0.001, sdf_grid=0.05, color_grid=0.05), w=dict(fs=5, center=200, tail=10, depth=0.1, color=5)),
    "rendering": dict(n_stratified=32, n_importance=8, perturb=True), "truncation": 0.06,
    "m_mask_mode": "original", "t_mask_mode": "original",
}


class TorchDraws:
    """
    The REFERENCE's random stream for the loop (r6): every draw the reference's tracker and mapper take from torch's global CPU generator, taken
    here from a torch.Generator in the same order and with the same shapes --
        tracking iteration     torch.randint(crop pixels, (n,))                                      src/common.py:116
                               torch.rand((R', S))  for the R' rays that passed the pre-filter        src/utils/Renderer.py:55
        keyframe selection     torch.randint(H * W, (50,)); torch.randperm(nonzero) while tracking back   src/Mapper.py:197-199,257
        pool of a frame        torch.randperm(H * W)[:10 %]                                          src/Mapper.py:335,518
        mapping iteration      torch.randint(P, (n * b,)) [+ torch.randint(P, (200 * 10,))]            src/common.py:155, src/Mapper.py:385-387
                               torch.rand((R', S)) [+ the two draws of the zero-depth branch]          src/utils/Renderer.py:55,117; src/common.py:64
    -- so that a SLAM(draws=TorchDraws(state=...)) run renders the same pixels with the same jitter as a reference run that started from
    that generator state.  With a draw source the drivers run eagerly (one host read of the pre-filter flags per iteration: the number of
    jitter rows is data-dependent in the reference); without one (default) every draw happens inside the kernels and the loops replay graphs.
    """

    def __init__(self, generator=None, state=None, seed=None):
        self.g = generator if generator is not None else torch.Generator()
        if state is not None:
            self.g.set_state(torch.as_tensor(state, dtype=torch.uint8))
        elif seed is not None:
            self.g.manual_seed(int(seed))

    def randint(self, high, n):
        return torch.randint(int(high), (int(n),), generator=self.g)

    def rand(self, rows, cols):
        return torch.rand((int(rows), int(cols)), generator=self.g)

    def randperm(self, n):
        return torch.randperm(int(n), generator=self.g)


def _jitter_rows(draws, valid, gd, S, n_strat, n_imp, perturb, has_zero):
    """the jitter of one iteration from the draw source, laid out for the kernels: t_rand [R, S] with the reference's row k in the k-th
    row that passed the pre-filter AND carries a depth (src/utils/Renderer.py:87-101), and -- windows with depth holes -- the zero-depth
    branch's two draws by COMPACTED row of !(depth > 0) (us_zero_depth_resample), the reference's row j in the j-th of them that passed"""
    v = valid.bool().cpu()
    d = gd.cpu() > 0
    R = v.shape[0]
    nz = v & d
    t_rand = torch.zeros((R, S))
    if perturb:
        t_rand[nz] = draws.rand(int(nz.sum()), S)
    zd = None
    z0 = v & ~d
    if has_zero:
        tu, u = torch.zeros((R, n_strat)), torch.zeros((R, n_imp))
        if bool(z0.any()):
            comp = torch.cumsum((~d).int(), 0) - 1                      # compacted row of every row without a depth
            rows = comp[z0]
            if perturb:
                tu[rows] = draws.rand(int(z0.sum()), n_strat)
            u[rows] = draws.rand(int(z0.sum()), n_imp)
        zd = (tu, u)
    elif bool(z0.any()):
        raise RuntimeError("draw replay: a ray without a depth in a window flagged has_zero_depth=False")
    return t_rand, zd


def keyframe_overlap(pts, keyframes_c2ws, H, W, fx, fy, cx, cy, edge=20):
    """Mapper.py:217-240: fraction of the sample points `pts` [M,3] that project inside each keyframe's image."""
    device = pts.device
    w2cs = rigid_inverse(keyframes_c2ws)                             # (Mapper.py:222 calls torch.inverse: the poses are rigid)
    ones = torch.ones_like(pts[..., :1])
    homo = torch.cat([pts, ones], dim=-1).reshape(1, -1, 4, 1).expand(w2cs.shape[0], -1, -1, -1)
    cam = (w2cs.unsqueeze(1).expand(-1, homo.shape[1], -1, -1) @ homo)[:, :, :3]
    K = torch.tensor([[fx, .0, cx], [.0, fy, cy], [.0, .0, 1.0]], device=device).reshape(3, 3)
    cam = cam.clone()
    cam[:, :, 0] *= -1
    uv = K @ cam
    z = uv[:, :, -1:] + 1e-5
    uv = uv[:, :, :2] / z
    mask = (uv[:, :, 0] < W - edge) * (uv[:, :, 0] > edge) * (uv[:, :, 1] < H - edge) * (uv[:, :, 1] > edge)
    mask = (mask & (z[:, :, 0] < 0)).squeeze(-1)
    return mask.sum(dim=1) / uv.shape[1]


def keyframe_selection_LC(num, idx, gt_color, gt_depth, c2w, num_keyframes, keyframe_list, estimate_c2w_list, cam, device,
                          tracking_back=False, activated_mapping_mode=True, LC=True, num_samples=8, num_rays=50, draws=None):
    """
    Mapper.keyframe_selection_LC (src/Mapper.py:177-274): indices into the keyframe list (excluding its last two entries) of the
    keyframes to optimise with the current view: ALL of them ("global"), from the loop partner on after a loop closure, or the
    num_keyframes best-overlapping ones while tracking back.  cam = (H, W, fx, fy, cx, cy).
    Returns (selected list, percent_inside, loop_closure).
    """
    H, W, fx, fy, cx, cy = cam
    if gt_depth.is_cuda and torch.is_tensor(estimate_c2w_list) and estimate_c2w_list.is_cuda:
        return _keyframe_selection_device(num, idx, gt_depth, c2w, num_keyframes, keyframe_list, estimate_c2w_list, cam, tracking_back,
                                          activated_mapping_mode, LC, num_samples, num_rays, draws)
    if draws is not None:
        raise RuntimeError("keyframe_selection_LC: a draw source is served by the device path (CUDA tensors)")
    rays_o, rays_d, gd, _ = get_samples(0, H, 0, W, num_rays, H, W, fx, fy, cx, cy, c2w.unsqueeze(0), gt_depth.unsqueeze(0),
                                        gt_color.unsqueeze(0), device)
    gd = gd.reshape(-1, 1)
    nz = gd[:, 0] > 0
    rays_o, rays_d, gd = rays_o[nz], rays_d[nz], gd[nz].repeat(1, num_samples)
    t_vals = torch.linspace(0., 1., steps=num_samples).to(device)
    z_vals = gd * 0.8 * (1. - t_vals) + (gd + 0.5) * t_vals
    pts = (rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]).reshape(-1, 3)
    if torch.is_tensor(estimate_c2w_list):                                               # (one gather instead of a slice per keyframe)
        kf_c2ws = estimate_c2w_list[torch.as_tensor(keyframe_list[:-2], dtype=torch.long, device=estimate_c2w_list.device)] if len(keyframe_list) > 2 \
            else estimate_c2w_list[:0]
    else:
        kf_c2ws = torch.stack([estimate_c2w_list[k] for k in keyframe_list], dim=0)[:-2]  # the last two are included anyway
    loop = False
    if kf_c2ws.shape[0] > 0:
        percent_inside = keyframe_overlap(pts, kf_c2ws, H, W, fx, fy, cx, cy)
        best = int(torch.argmax(percent_inside))
        idx1 = keyframe_list[best]
        if (idx - idx1) > 100 and LC and float(percent_inside[best]) > 0.95:
            selected = list(range(0, num))[best:]                                        # only the frames of this loop
            loop = True
        else:
            selected = list(range(0, num))
    else:
        percent_inside = torch.zeros(0, device=device)
        selected = list(range(0, num))
    if tracking_back and activated_mapping_mode:                                         # local BA while tracking back (:253-272)
        sel = torch.nonzero(percent_inside).squeeze(-1)
        sel = sel[torch.randperm(sel.shape[0])[:num_keyframes]]                          # (the draw is made, then overridden)
        selected = [int(k) for k in sel.cpu().numpy()]
        if percent_inside.shape[0] > 0:                                                  # the num_keyframes best-overlapping ones
            order = sorted(range(len(percent_inside)), key=lambda k: float(percent_inside[k]), reverse=True)
            selected = [k for k in order if percent_inside[k] > 0.00][:num_keyframes]
    return selected, percent_inside, loop


def rigid_inverse(c2ws):
    """inverse of [n,4,4] rigid transforms in closed form: [R^T | -R^T t] (torch.inverse initialises a solver library on its first call:
    0.3 s in the middle of the sequence)"""
    Rt = c2ws[:, :3, :3].transpose(1, 2)
    out = torch.zeros_like(c2ws)
    out[:, :3, :3] = Rt
    out[:, :3, 3] = -(Rt @ c2ws[:, :3, 3:4])[:, :, 0]
    out[:, 3, 3] = 1.0
    return out


def _keyframe_selection_device(num, idx, gt_depth, c2w, num_keyframes, keyframe_list, estimate_c2w_list, cam, tracking_back,
                               activated_mapping_mode, LC, num_samples, num_rays, draws=None):
    """keyframe_selection_LC for device tensors: the pixel draw, ONE launch for the overlap of every keyframe (us_keyframe_overlap: rays,
    points, rigid inverse and projection on the fly), ONE read of the K shares; the selection rules on the host as above"""
    from . import _lib as L
    H, W, fx, fy, cx, cy = cam
    dev = gt_depth.device
    K = len(keyframe_list) - 2
    loop, selected = False, list(range(0, num))
    pct = torch.zeros(max(K, 0), device=dev)
    vals = []
    pix_drawn = draws.randint(H * W, num_rays).to(dev) if draws is not None else None    # (the reference draws whatever K is: Mapper.py:197)
    if K > 0:
        pix = pix_drawn if draws is not None else torch.randint(H * W, (num_rays,), device=dev)   # common.py:116: the draw of get_samples
        kf = torch.as_tensor(keyframe_list[:K], dtype=torch.int64, device=dev)
        L.check(L.lib().us_keyframe_overlap(L.ptr(L.f32(c2w)), L.ptr(L.f32(gt_depth)), L.ptr(pix), num_rays, num_samples, L.host_floats([fx, fy, cx, cy]),
                                            H, W, 20, L.ptr(L.f32(estimate_c2w_list)), L.ptr(kf), K, L.ptr(pct), L.stream()), "us_keyframe_overlap")
        vals = pct.tolist()                                                              # the one host read of the selection
        best = max(range(K), key=lambda k: (vals[k], -k))                                # argmax, the first of equal maxima
        if vals[best] > 0.95 and (idx - keyframe_list[best]) > 100 and LC:
            selected, loop = list(range(0, num))[best:], True
    if tracking_back and activated_mapping_mode:
        selected = []
        if draws is not None:
            draws.randperm(sum(1 for x in vals if x != 0))                               # Mapper.py:256-257: drawn, then overridden by the ranking
        if K > 0:
            order = sorted(range(K), key=lambda k: vals[k], reverse=True)
            selected = [k for k in order if vals[k] > 0.00][:num_keyframes]
    return selected, pct, loop


class Mapper:
    """Mapper.run body for one frame + optimize_mapping (src/Mapper.py:276-459,461-545).

    The keyframes' pixel pools live in a KeyframeArena (one row per keyframe, written once); a mapping window is a list of rows.  The
    15 / 20 / 30 iterations of a mapped frame replay ONE captured hipGraph per kind of window -- (joint_opt, extra rays, depth holes,
    lr factor): at most a handful per run, captured ahead by prewarm() -- whatever the number of frames in the window (ArenaWindow: the
    window's shape is read on the device).  cfg['mapping']['graph_replay'] False runs the same windows eagerly."""

    def __init__(self, slam):
        self.s = slam
        c = slam.cfg["mapping"]
        r = slam.cfg["rendering"]
        self.c = c
        self.step = MapStep(slam.es, slam.ec, slam.decoders, slam.bound, r["n_stratified"], r["n_importance"], slam.cfg["truncation"],
                            c["w"], c["lr"], mask_mode=slam.cfg["m_mask_mode"], perturb=r["perturb"], max_rays=c["pixels"] + 2000)
        self.keyframe_list, self.keyframe_dict = [], []
        self.init_phase, self.LC_cnt, self.joint_opt = True, 0, False
        H, W = slam.cam[0], slam.cam[1]
        n_kf = int(c.get("arena_keyframes", len(slam.frames) // max(1, c["keyframe_every"]) + 8))
        self.arena = KeyframeArena(n_kf + 1, int(H * W * 0.1), slam.device)
        self.kf_c2w = torch.zeros((self.arena.K, 4, 4), device=slam.device)             # est_c2w per arena row (src/Mapper.py:447-457 writes them back)
        self._wins = {}
        self.kind_counts = {}
        self.cur_has_zero = False
        self.timing = None

    # ---------------------------------------------------------------------------------------------- pools
    def _pool_into(self, row, color, depth, rays_d):
        """10 % of the pixels of a frame as its sampling pool (Mapper.py:329-337,516-523), written into arena row `row`; returns whether the
        pool holds pixels without a depth (one host read per pool: it picks the window's graph)"""
        d = self.s.draws
        if d is not None:                                                               # the reference's cut: torch.randperm(H * W)[:10 %] (Mapper.py:335,518)
            ind = d.randperm(depth.numel())[:self.arena.P].to(depth.device)
            dp = depth.reshape(-1)[ind]
            self.arena.put(row, color.reshape(-1, 3)[ind], dp, rays_d.reshape(-1, 3)[ind])
            return bool((dp <= 0).any())
        self._pool_count = getattr(self, "_pool_count", 0) + 1
        seed = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + self._pool_count) & (2 ** 64 - 1)
        return bool(int(self.arena.cut(row, color, depth, rays_d, seed)))

    def _window(self, joint, extra_on, has_zero, lr_factor):
        """the ArenaWindow (and its graph) for this kind of mapped frame"""
        c = self.c
        if self.arena.generation != getattr(self, "_gen", 0):                           # the arena grew: captured graphs hold old addresses
            self._wins.clear(); self._gen = self.arena.generation
            kc = torch.zeros((self.arena.K, 4, 4), device=self.s.device); kc[:self.kf_c2w.shape[0]] = self.kf_c2w; self.kf_c2w = kc
        key = (bool(joint), bool(extra_on), bool(has_zero), float(lr_factor))
        win = self._wins.get(key)
        if win is None:
            self.step.reset_optimizer(lr_factor)                                        # the learning rates the graph records
            win = ArenaWindow(self.step, self.arena, c["pixels"], 2000 if extra_on else 0, joint_opt=joint, cam_lr=c["joint_opt_cam_lr"],
                              has_zero_depth=has_zero)
            if bool(c.get("graph_replay", True)) and self.s.draws is None:
                win.capture()
            self._wins[key] = win
        return win

    def prewarm(self, kinds=None):
        """capture the graphs of the window kinds a run meets -- first keyframes (poses fixed), joint_opt, joint_opt with the extra rays of
        the newest frames -- before the first frame, with the arena's placeholder pixels; also allocates the joint_opt scratch.  Nothing
        the model, the optimiser or the sequence sees is changed."""
        c = self.c
        if kinds is None:
            kinds = [(False, False, False, c["lr_factor"]), (c["joint_opt"], False, False, c["lr_factor"]), (c["joint_opt"], True, False, c["lr_factor"])]
        for k in kinds:
            self._window(*k)
        self.step.reset_optimizer(1.0)
        # keyframe selection on a stand-in list: its first call loads a dozen torch kernels and the BLAS library behind the batched
        # projection (0.25 s when that happened at the fourth keyframe)
        s = self.s
        H, W = s.cam[0], s.cam[1]
        eye = torch.eye(4, device=s.device)
        est = eye[None].repeat(5, 1, 1)
        keyframe_selection_LC(3, 4, torch.zeros((H, W, 3), device=s.device), torch.ones((H, W), device=s.device), eye, c["mapping_window_size"] - 1,
                              [0, 1, 2, 3, 4], est, s.cam, s.device, False, s.cfg["tracking"]["activated_mapping_mode"], c["LC"])
        keyframe_selection_LC(3, 4, torch.zeros((H, W, 3), device=s.device), torch.ones((H, W), device=s.device), eye, c["mapping_window_size"] - 1,
                              [0, 1, 2, 3, 4], est, s.cam, s.device, True, s.cfg["tracking"]["activated_mapping_mode"], c["LC"])
        self._pool_into(0, torch.zeros((H, W, 3), device=s.device), torch.ones((H, W), device=s.device), torch.zeros((H, W, 3), device=s.device))
        torch.cuda.synchronize()

    # ---------------------------------------------------------------------------------------------- one mapped frame
    def _mark(self, name):
        """development: with self.timing = {} set, wall time (device synchronised) of the phases of a mapped frame"""
        if self.timing is not None:
            import time
            torch.cuda.synchronize()
            t = time.perf_counter()
            self.timing.setdefault(name, []).append(1e3 * (t - self._t_mark))
            self._t_mark = t

    def optimize_mapping(self, iters, lr_factor, idx, cur_color, cur_depth, cur_c2w, cur_rays_d):
        s, c, dev = self.s, self.c, self.s.device
        kd, kl = self.keyframe_dict, self.keyframe_list
        if self.timing is not None:
            import time
            torch.cuda.synchronize(); self._t_mark = time.perf_counter()
        if len(kd) == 0:
            optimize_frame = []
        else:
            optimize_frame, _, loop = keyframe_selection_LC(len(kd) - 2, idx, cur_color, cur_depth, cur_c2w, c["mapping_window_size"] - 1,
                                                            kl, s.estimate_c2w_list, s.cam, dev, s.tracking_back,
                                                            s.cfg["tracking"]["activated_mapping_mode"], c["LC"], draws=s.draws)
            self.LC_cnt += int(loop)
        if len(kl) > 1:
            optimize_frame = sorted(optimize_frame + [len(kl) - 1] + [len(kl) - 2])
        optimize_frame += [-1]                                                          # -1 = the current frame
        self._mark("keyframe selection")
        b = len(optimize_frame)
        pixs_per_image = c["pixels"] // b
        # the window = arena rows: the selected keyframes' (written when they were made), row 0 = the pool of the frame being mapped
        self.cur_has_zero = self._pool_into(0, cur_color, cur_depth, cur_rays_d)
        rows = [kd[f]["row"] if f != -1 else 0 for f in optimize_frame]
        self.kf_c2w[0] = cur_c2w
        c2ws = self.kf_c2w[torch.as_tensor(rows, device=dev)]
        has_zero = self.cur_has_zero or any(kd[f]["has_zero"] for f in optimize_frame if f != -1)
        joint = self.joint_opt and b > 1
        extra = (10, 200) if (not s.tracking_back and len(kl) > 20 and c.get("extra_rays", True)) else None   # extra rays from the newest frames (:381-390)
        self._mark("pool of the frame + window rows")
        win = self._window(joint, extra is not None, has_zero, lr_factor)
        kind = (bool(joint), extra is not None, bool(has_zero), float(lr_factor))
        self.kind_counts[kind] = self.kind_counts.get(kind, 0) + 1                    # how often each kind of window was mapped (tests, logs)
        self.step.reset_optimizer(lr_factor)                                            # a fresh Adam per mapped frame (:358-364)
        win.bind(rows, c2ws, pixs_per_image, extra)
        self._mark("optimiser reset + bind")
        # the loop of :366-445: pose -> rays, render, loss, backward, pose step and Adam are HIP launches on static buffers, one graph
        step = win.replay if win._graph is not None else win.iterate
        if s.draws is not None:
            step = lambda: self._iterate_drawn(win)
        for _ in range(int(iters)):
            step()
        s.history["mapped"].append(dict(idx=int(idx), iters=int(iters), frames=list(optimize_frame), joint=bool(joint), lr_factor=float(lr_factor)))
        self._mark("iterations")
        if joint:
            opt = win.c2ws()                                                            # put the updated camera poses back (:447-457)
            self.kf_c2w[torch.as_tensor(rows[1:], device=dev)] = opt[1:]
            cur_c2w = opt[-1]
        self._mark("pose write-back")
        return cur_c2w

    def _iterate_drawn(self, win):
        """one mapping iteration with the pixels and the jitter of the draw source (TorchDraws): src/common.py:155, src/Mapper.py:385-387,
        src/utils/Renderer.py:55 in the reference's order"""
        d, st, dev = self.s.draws, self.step, self.s.device
        ia = d.randint(win.P, win.b * win.n_per).to(dev)
        ib = d.randint(win.P, win.extra[0] * win.extra[1]).to(dev) if win.extra else None
        win.draw(ia, ib)
        valid, gd = win.probe_valid()
        t_rand, zd = _jitter_rows(d, valid, gd, st.S, st.n_strat, st.n_imp, st.perturb, win.has_zero)
        zd = None if zd is None else (zd[0].to(dev), zd[1].to(dev))
        loss = win.iterate(ia, ib, t_rand=t_rand.to(dev), zero_depth_draws=zd)
        self.s.history["losses"].append(float(loss))
        return loss

    def map_frame(self, idx, color, depth, gt_c2w, rays_d):
        """one pass of the Mapper.run loop body for frame idx (Mapper.py:494-533)"""
        s, c = self.s, self.c
        cur_c2w = s.estimate_c2w_list[idx]
        lr_factor = c["lr_first_factor"] if self.init_phase else c["lr_factor"]
        iters = c["iters_first"] if self.init_phase else s.m_iters
        self.joint_opt = (len(self.keyframe_list) > 4) and c["joint_opt"]
        cur_c2w = self.optimize_mapping(iters, lr_factor, idx, color, depth, cur_c2w, rays_d)
        if self.joint_opt:
            s.estimate_c2w_list[idx] = cur_c2w
        if idx % c["keyframe_every"] == 0 or s.tracking_back:
            self.keyframe_list.append(idx)
            row = self.arena.alloc()
            if self.arena.generation != getattr(self, "_gen", 0):
                self._window(False, False, False, c["lr_factor"])                       # (rebuilds kf_c2w and the graphs for the grown arena)
            hz = self._pool_into(row, color, depth, rays_d)
            self.kf_c2w[row] = cur_c2w
            self.keyframe_dict.append({"gt_c2w": gt_c2w, "idx": idx, "row": row, "has_zero": hz})
            if self.timing is not None:
                self._mark("new keyframe's pool")
        self.init_phase = False

    def keyframe_pose(self, k):
        """est_c2w of keyframe k (the reference keeps it in keyframe_dict[k]['est_c2w'])"""
        return self.kf_c2w[self.keyframe_dict[k]["row"]]


class Tracker:
    """Tracker.run body for one frame (src/Tracker.py:296-361)"""

    def __init__(self, slam):
        self.s = slam
        c, r = slam.cfg["tracking"], slam.cfg["rendering"]
        self.c = c
        self.step = TrackStep(slam.es, slam.ec, slam.decoders, slam.bound, r["n_stratified"], r["n_importance"], slam.cfg["truncation"],
                              c["w"], mask_mode=slam.cfg["t_mask_mode"], perturb=r["perturb"], max_rays=c["pixels"])
        self.num_cam_iters = c["iters"]
        self.rendered_weight = {}
        self._graphs, self._graphs_gen = {}, 0
        self.params_stale = True                                                        # set by SLAM.run after every mapping pass

    def track_frame(self, idx, color, depth):
        s, c, dev = self.s, self.c, self.s.device
        H, W, fx, fy, cx, cy = s.cam
        pre_c2w = s.estimate_c2w_list[idx - 1]
        if c["const_speed_assumption"] and idx - 2 >= 0:                                # linear prediction (:317-320)
            cam_pose = predict_cam_pose(s.estimate_c2w_list[idx - 2], pre_c2w)
        else:
            cam_pose = matrix_to_cam_pose(pre_c2w.unsqueeze(0))
        begin = lambda: self.step.begin_frame(cam_pose[0], color, depth, c["lr_T"], c["lr_R"], H, W, fx, fy, cx, cy, c["ignore_edge_H"],
                                              c["ignore_edge_W"], betas=(0.5, 0.999), refresh=self.params_stale)
        begin()                                                                         # (update_params_from_mapping, :302: after a mapping pass)
        self.params_stale = False
        # The iteration is nine short launches: issued one by one from Python the loop is host-bound (2.2 ms per frame at Replica's
        # settings for 1.1 ms of kernels).  Every buffer of TrackStep is static, the pixel draw and the jitter happen in the kernels,
        # and the optimiser's state lives on the device, so captured graphs serve every frame: one graph per RUN LENGTH -- the k
        # iterations up to the loop's one host decision (7, or 8 / 15 while tracking back) in one launch, single iterations behind it.
        replay = bool(c.get("graph_replay", True))
        n_pix = c["pixels"]

        def run(k):
            if not replay:
                for _ in range(k):
                    out = self.step.iterate_fused(n_pix)
                return out
            if self._graphs and self._graphs_gen != self.step.generation:
                self._graphs.clear()                                                    # TrackStep reallocated: the graphs hold old addresses
            if k not in self._graphs:
                self._capture(k, begin)
            return self._graphs[k].replay()

        if s.draws is not None:
            run = lambda k: [self._iterate_drawn(n_pix) for _ in range(k)][-1]
        it = 0
        while it < self.num_cam_iters:                                                  # re-read: the count may double mid-frame
            # the minimum-loss candidate (:346-348) is kept by the pose step's launch: step.min_loss / step.best_pose, on the device
            k = max(self.num_cam_iters - 1 - it, 1)                                     # iterations up to the check of :352, or one behind it
            loss, unc, valid = run(k)
            it += k
            if it == self.num_cam_iters - 1:                                            # (:352-364)
                w = self.step.mean_uncertainty(unc, valid).clone()
                self.rendered_weight[idx] = w
                if c["activated_mapping_mode"] and float(w) > c["uncertainty_ts"]:
                    self.num_cam_iters = c["iters"] * 2
                    s.m_iters, s.tracking_back = s.cfg["mapping"]["iters"] * 2, True
                else:
                    self.num_cam_iters = c["iters"]
                    s.m_iters, s.tracking_back = s.cfg["mapping"]["iters"], False
        s.history["track_iters"][idx] = it
        return cam_pose_to_matrix(self.step.best_pose.reshape(1, 7))[0]

    def _iterate_drawn(self, n_pix):
        """one tracking iteration with the pixels and the jitter of the draw source (TorchDraws): src/common.py:116, src/utils/Renderer.py:55"""
        d, st, dev = self.s.draws, self.step, self.s.device
        H, W, eh, ew = st.frame
        pix = d.randint((H - 2 * eh) * (W - 2 * ew), n_pix).to(dev)
        v = st.probe_valid(n_pix, pix).bool().cpu()
        t_rand = torch.zeros((n_pix, st.S))
        if st.perturb:
            t_rand[v] = d.rand(int(v.sum()), st.S)
        out = st.iterate_fused(n_pix, t_rand=t_rand.to(dev), indices=pix)
        self.s.history["losses"].append(float(out[0]))
        return out


    def _capture(self, k, begin):
        """the graph of a run of k tracking iterations (a capture records, it does not execute); begin(): the frame's set-up, run again after
        the one eager iteration that allocates the lazy buffers"""
        from .graph import CapturedIteration
        n_pix = self.c["pixels"]
        if not self._graphs:
            self.step.iterate_fused(n_pix)
            begin()

        def body():
            for _ in range(k):
                out = self.step.iterate_fused(n_pix)
            return out
        self._graphs[k] = CapturedIteration(body, warmup=0)
        self._graphs_gen = self.step.generation

    def prewarm(self, color, depth, c2w):
        """capture the run lengths a sequence meets (iters - 1 up to the loop's host decision, 1 behind it, and their doubled forms while
        tracking back) on a stand-in frame, so that no tracked frame pays for a capture"""
        s, c = self.s, self.c
        if not bool(c.get("graph_replay", True)):
            return
        H, W, fx, fy, cx, cy = s.cam
        pose = matrix_to_cam_pose(c2w.unsqueeze(0))
        begin = lambda: self.step.begin_frame(pose[0], color, depth, c["lr_T"], c["lr_R"], H, W, fx, fy, cx, cy, c["ignore_edge_H"],
                                              c["ignore_edge_W"], betas=(0.5, 0.999), refresh=True)
        begin()
        # the loop's run lengths (track_frame): k = max(n - 1 - it, 1) with n = iters or 2 * iters, n possibly switching at it = n - 1
        n1, n2 = c["iters"], 2 * c["iters"]
        for k in sorted({max(n1 - 1, 1), 1, max(n2 - 1, 1), max(n2 - 1 - (n1 - 1), 1)}):
            if k not in self._graphs:
                self._capture(k, begin)
        self.params_stale = True


class SLAM:
    """
    Sequential tracking + mapping over a frame source (items: idx, color [H,W,3], depth [H,W], gt_c2w [4,4], rays_d [H,W,3]).
    hash grids / decoders: the modules the reference builds at src/UNISLAM.py:241-259.
    """

    def __init__(self, frames, cam, hash_grid_sdf, hash_grid_color, decoders, bound, cfg=None, draws=None):
        """draws: None (default: pixels and jitter are drawn inside the kernels, the loops replay captured graphs) | a TorchDraws: the
        reference's random stream, consumed as its loop consumes it (eager iterations; for draw-for-draw comparisons with a reference run)"""
        import copy
        self.draws = draws
        self.history = {"track_iters": {}, "tracking_back": {}, "mapped": [], "losses": []}          # the loop's decisions, for logs and tests
        self.frames, self.cam = frames, cam
        self.cfg = copy.deepcopy(DEFAULTS)
        for k, v in (cfg or {}).items():
            if isinstance(v, dict):
                self.cfg[k].update(v)
            else:
                self.cfg[k] = v
        self.es, self.ec, self.decoders, self.bound = hash_grid_sdf, hash_grid_color, decoders, bound
        self.device = hash_grid_sdf.params.device
        n = len(frames)
        self.estimate_c2w_list = torch.zeros((n, 4, 4), device=self.device)
        self.gt_c2w_list = torch.zeros((n, 4, 4), device=self.device)
        self.m_iters, self.tracking_back = self.cfg["mapping"]["iters"], False
        self.mapper, self.tracker = Mapper(self), Tracker(self)

    def run(self, n_frames=None, log=None, start=0, total=None):
        """frames [start, n_frames) of the sequence; total: the sequence's length where it is not n_frames (its last frame is always
        mapped, src/Mapper.py:487-488).  start > 0 continues a run whose state the caller has put in place (estimate_c2w_list[:start], the
        mapper's keyframes, the iteration counts)."""
        every = self.cfg["mapping"]["every_frame"]
        n = len(self.frames) if n_frames is None else n_frames
        last = (n if total is None else int(total)) - 1
        if n > 0 and self.cfg.get("prewarm", True) and self.draws is None and not getattr(self, "_prewarmed", False):
            # everything a frame would otherwise pay for once, somewhere in the sequence: graph captures of both loops, the joint_opt
            # scratch, first launches of every kernel -- on a stand-in (frame 0), before the first frame is timed by anyone
            _, color, depth, gt_c2w, rays_d = self.frames[0]
            color, depth, gt_c2w = (t.to(self.device) for t in (color, depth, gt_c2w))
            hz = bool((depth <= 0).any())
            c = self.cfg["mapping"]
            if bool(c.get("graph_replay", True)):
                self.mapper.prewarm([(False, False, hz, c["lr_factor"]), (c["joint_opt"], False, hz, c["lr_factor"]),
                                     (c["joint_opt"], True, hz, c["lr_factor"])])
            self.tracker.prewarm(color, depth, gt_c2w)
            self._prewarmed = True
        for idx in range(int(start), n):
            _, color, depth, gt_c2w, rays_d = self.frames[idx]
            color, depth, gt_c2w, rays_d = (t.to(self.device, non_blocking=True) for t in (color, depth, gt_c2w, rays_d))   # disk readers yield CPU tensors (UNISLAM/Tracker.py:303-306)
            self.gt_c2w_list[idx] = gt_c2w
            if idx == 0:
                self.estimate_c2w_list[0] = gt_c2w                                       # the first pose is given (Mapper.py:479)
            else:
                self.estimate_c2w_list[idx] = self.tracker.track_frame(idx, color, depth)
            self.history["tracking_back"][idx] = bool(self.tracking_back)
            if idx % every == 0 or self.tracking_back or idx == last:                   # Mapper.py:487-493
                self.mapper.map_frame(idx, color, depth, gt_c2w, rays_d)
                self.tracker.params_stale = True
            if log is not None:
                log(idx, self)
        return self.estimate_c2w_list[:n]

    def evaluate(self, n=None, pose_alignment=False):
        """the reference's ATE report (src/tools/eval_ate.py:270-281): Horn-aligned translational error in cm -> (errors [n], results)"""
        from .eval_ate import pose_evaluation
        n = self.estimate_c2w_list.shape[0] if n is None else n
        return pose_evaluation(self.gt_c2w_list[:n].cpu(), self.estimate_c2w_list[:n].cpu(), scale=1.0, pose_alignment=pose_alignment)

    def evaluate_rendering(self, n=None, stride=5, truncation=None):
        """the reference's render-quality report (src/tools/eval_recon.py:235-307): PSNR and depth L1 of frames re-rendered at
        their estimated poses -> {"avg_psnr", "depth_l1_render", "frames"}"""
        from .eval_render import eval_rendering
        from .renderer import Renderer
        H, W, fx, fy, cx, cy = self.cam
        r = self.cfg["rendering"] if "rendering" in self.cfg else {"n_stratified": 32, "n_importance": 8}
        rend = Renderer({"rendering": {"perturb": False, "n_stratified": r["n_stratified"], "n_importance": r["n_importance"]}, "scale": 1,
                         "grid_mode": "hash_grid"},
                        type("U", (), dict(bound=self.bound, device=self.device, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy))())
        n = self.estimate_c2w_list.shape[0] if n is None else n
        tr = self.cfg["truncation"] if truncation is None else truncation
        return eval_rendering(n, self.frames, self.estimate_c2w_list, rend, ([self.es], [self.ec]), self.decoders, tr, self.device, stride)

    def ate_rmse(self, n=None):
        """translation RMSE of the estimated trajectory against the given one, no alignment (frame 0 is shared)"""
        n = self.estimate_c2w_list.shape[0] if n is None else n
        d = self.estimate_c2w_list[:n, :3, 3] - self.gt_c2w_list[:n, :3, 3]
        return float(d.pow(2).sum(-1).mean().sqrt())
