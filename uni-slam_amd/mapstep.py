"""
MapStep -- the body of the reference's mapping hot loop (src/Mapper.py:366-445: sample -> render -> masked loss ->
backward -> Adam) as ONE straight-line sequence of HIP kernel launches on preallocated buffers: no autograd graph,
no boolean-mask compaction, no host synchronisation.  Results are the same as driving Renderer / Decoders / losses
through autograd (tests/test_gpu_step.py holds the two paths against each other and against the oracle).

Parameter storage.  All trainable parameters live in ONE flat fp32 buffer

    [ sdf decoder | colour decoder | beta | pad ][ sdf hash table ][ colour hash table ]

laid out so that each decoder segment is directly the flat vector the fused MLP kernel consumes (nn.Linear weights are
re-pointed to views of it; the last matrix is zero-padded to 16 rows) and each table segment is the encoder's
`params`.  The modules keep working (state_dict, deepcopy, Tracker reading the shared tables); gradients live in a
second flat buffer of the same layout, which is what the single RCCL all-reduce per optimiser step moves
(BASELINE.json north_star), and Adam runs over the three learning-rate groups of Mapper.create_optimizer
(src/Mapper.py:111-139).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L
from .common import bound_host
from .decoders import Decoders
from .hashgrid import HashGridEncoding
from .network import make_mlp_desc, mlp_n_params
from .dist import dp_iterate


def _align(n, a=2048):     # 2048 floats: 16-byte alignment and equal shards for 1, 2, 4, 8 ... ranks (dist: sharded Adam)
    return (n + a - 1) // a * a



class MapStep:
    # flags of the table gradient's bin policy: what the constructor / enable_grad_image asked for, plus "no split bins" while the tables'
    # optimiser step rides in the accumulate pass (fuse_adam): scan call and gradient call must agree
    @property
    def _det(self):
        return self._det_base | (L.US_GRID_BWD_DETERMINISTIC if getattr(self, "_fuse_adam", False) else 0)

    @_det.setter
    def _det(self, v):
        self._det_base = int(v)

    # fuse_adam (EXPERIMENTS build only: us_hashgrid_bwd_joint_adam, include/unislam_hip_experiments.h): see __init__
    @property
    def fuse_adam(self):
        return self._fuse_adam

    @fuse_adam.setter
    def fuse_adam(self, on):
        on = bool(on)
        if on and not L.has_experiments():
            raise L.UniSlamHipError("MapStep.fuse_adam: us_hashgrid_bwd_joint_adam belongs to the experiments build (tools/build_experiments.sh); "
                                    "it was measured slower than the separate optimiser pass (DESIGN.md 9)")
        if on != getattr(self, "_fuse_adam", False):
            # the bin policy changes with it (no split bins): counts / scans of a pending forward pass and captured graphs are void
            self._jcounted = False
            self._graph = None
            self.generation = getattr(self, "generation", 0) + 1
        self._fuse_adam = on

    def __init__(self, hash_grid_sdf, hash_grid_color, decoders, bound, n_stratified, n_importance, truncation,
                 weights, lr, mask_mode="original", perturb=True, max_rays=4096, group=None, bwd_mode=-1, overlap=None, grad_comm=None, sharded_adam=False,
                 joint=None, deterministic=False, max_workspace_bytes=4 << 30, dp_mode="local_fast"):
        """
        hash_grid_sdf / hash_grid_color: HashGridEncoding;  decoders: Decoders (either parameterisation);
        weights: dict(fs, center, tail, color, depth)   (cfg['mapping']['w_*'], src/Mapper.py:63-67);
        lr: dict(decoders, sdf_grid, color_grid)        (cfg['mapping']['lr'], src/Mapper.py:123-126);
        group: None | True (default process group) | a torch.distributed group -> data-parallel over ranks.
        joint: encode both grids in one launch and form both table gradients in one binned pass (us_hashgrid_fwd_joint /
                 us_hashgrid_bwd_joint: the grids share cells, runs and hashes).  Default: yes when the pair of grids qualifies (with a
                 process group: see dp_mode -- the accumulate pass is then split per grid, the colour table first).
                 Measured at 4096 x 64 (room0 tables): the table gradient is 262 us for both grids against 157 + 147; render-only calls
                 (backward_follows=False) run the joint encoder without counts and both decoders in one launch.
        overlap: use side streams -- one-grid kernels: the sdf branch (encode, decode and their backward) beside the colour branch; joint
                 kernels: the binning's scans and the small reductions beside the main chain (the decoders stay on the main stream: a
                 dependency across queues costs 10-14 us, and the decoder kernels fill the chip on their own).  Default: see dp_mode.
        max_workspace_bytes: budget of ONE scratch set of the table gradient (the joint path has one; the one-grid path two, one per
                 branch): a batch that would need more is walked in ranges of rays.  Default 4 GiB.
        """
        assert isinstance(hash_grid_sdf, HashGridEncoding) and isinstance(hash_grid_color, HashGridEncoding)
        assert isinstance(decoders, Decoders)
        self.es, self.ec, self.dec = hash_grid_sdf, hash_grid_color, decoders
        # dp_mode (with a process group, when joint / overlap are left to default): "local_fast" -- the single-process kernels (joint
        # encoder + joint record pass, side streams for the scans and small reductions) with the accumulate pass split per grid, colour
        # first, so that the colour table's all-reduce travels behind the sdf table's accumulate pass; "colour_first" -- the one-grid kernels
        # on one stream, colour branch before sdf branch: a longer cover for that all-reduce (the whole sdf branch) at a higher cost per rank.
        if dp_mode not in ("local_fast", "colour_first"):
            raise L.UniSlamHipError(f"MapStep: dp_mode {dp_mode!r} not in ('local_fast', 'colour_first')")
        self.dp_mode = dp_mode
        fast_default = group is None or dp_mode == "local_fast"
        self.overlap, self.side, self.scan_stream = fast_default if overlap is None else bool(overlap), None, None
        self.decoder_pair = True        # the two decoders, where they have one shape, as one launch each way (a launch costs ~5 us whatever it computes)
        self._dec_grad_clean = False
        self._step_advanced = False
        self._scan_pending = self._side_pending = False          # work queued on the scan / side stream since its last join
        self.one_launch_adam = True                               # single process: decoder group + tables in ONE optimiser launch (us_adam_step_model)
        self.adam_in_parts = True                                # data-parallel: the colour table's optimiser pass ahead of the rest (dist.dp_iterate)
        self._grad_bf16_from = None                              # dist.GradComm (bf16 payload): first flat index whose gradient lives in self._grad_bf16
        self._grad_image = False                                 # ... and the accumulate pass writes the colour table's part of it (enable_grad_image)
        self._grad_image_written = False
        # store_dydx: the joint encoder of a forward(backward_follows=True) also leaves d(features)/d(position) (us_hashgrid_fwd_joint_dydx),
        # and backward(ray_grads=True) contracts it (us_hashgrid_dydx_rays) instead of gathering the tables a second time.  Set by
        # window.MapWindow for the iterations that optimise camera poses (src/Mapper.py:372-376); costs 2 x 24 B per point and level.
        self.store_dydx, self._dydx_valid, self.dydx_s, self.dydx_c = False, False, None, None
        self._joint_wanted = fast_default if joint is None else bool(joint)
        # deterministic: hot bins of the table gradient are not split over workgroups (US_GRID_BWD_DETERMINISTIC): no float atomics, the
        # gradients repeat bit for bit from run to run (the decoder gradients already do: per-workgroup partials, fixed-order sums)
        self._det = L.US_GRID_BWD_DETERMINISTIC if deterministic else 0
        # fuse_adam (r5; EXPERIMENTS build, single process, joint kernels): iterate() applies the tables' Adam step INSIDE the
        # accumulate pass's sweep (us_hashgrid_bwd_joint_adam) -- no gradient table written, none read back, the optimiser launch shrinks to
        # the decoders' (and poses') group.  Every entry is then written exactly once, by one workgroup: bins are not split
        # (US_GRID_BWD_DETERMINISTIC, measured at no cost: 0.4886 against 0.4884 ms at 4096 x 64, 0.508 against 0.507 with all rays from ONE
        # camera).  Same bits as the separate pass (tests/test_gpu_window.py).  OFF by default -- measured, MI355X, 4096 x 64 replayed: the
        # accumulate pass grows from 78 to 121 us (the sweep's reads of p, m, v sit between two barriers of every item, their latency
        # exposed; no registers left to prefetch them: 63 of 64), the optimiser launch falls from 55 to ~8 us: 0.4926 ms against 0.4884
        # for the separate streaming pass, which moves its 362 MB at 6.5 TB/s with the decoders' group riding along; with the small launch
        # on the side stream beside the table gradient: 0.505 (a queue crossing).  keep_table_grad: also write the gradient tables.
        self.fuse_adam = False
        self.keep_table_grad = False
        self._tables_stepped = False
        # budget of the table gradient's scratch (it is sized for the worst case, 8 records per point and level: 3 KB per point for
        # both grids).  A batch that would need more is walked in ranges of rays (us_hashgrid_bwd_*_range), the first range writing
        # the gradient tables, the others adding: 4096 x 64 needs 0.8 GB, the 32 768-ray sweep point 6.6 GB -> two ranges.
        self.max_ws = int(max_workspace_bytes)
        self.count_in_forward, self._counted = True, False
        self.grad_comm = grad_comm      # None/"fp32" | "bf16" | "bf16_colour": payload type of the gradient all-reduce (dist.GradComm)
        self.sharded_adam = bool(sharded_adam)   # dist.dp_iterate: reduce-scatter, Adam on this rank's shard, all-gather
        if self.sharded_adam and grad_comm in ("bf16", "bf16_colour", torch.bfloat16):
            raise L.UniSlamHipError("MapStep: grad_comm='bf16' and sharded_adam=True are exclusive (the reduce-scatter runs in place on "
                                    "the fp32 gradient buffer); choose one")
        self.rng_seed, self.rng_calls = int(torch.initial_seed()) & (2 ** 63 - 1), 0    # in-kernel jitter / pixel-draw generator
        if group is not None:
            # Data-parallel ranks usually share the torch seed (identical initial replicas): without a rank in the generator's seed they
            # would draw the same pool pixels and the same jitter, and the all-reduce would average W copies of ONE batch.
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                r = dist.get_rank(None if group is True else group)
                x = (r + 1) * 0x9E3779B97F4A7C15 & (2 ** 64 - 1)                    # splitmix64 finaliser of the rank
                x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
                x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
                self.rng_seed = (self.rng_seed ^ x ^ (x >> 31)) & (2 ** 63 - 1)
        self.zd_rows = None                                                                # scratch of the zero-depth branch, on first use
        dev = hash_grid_sdf.params.device
        if dev.type != "cuda":
            raise L.UniSlamHipError("MapStep needs the model on the GPU")
        self.device = dev
        self.S = n_stratified + n_importance
        self.n_strat, self.n_imp = n_stratified, n_importance
        self.truncation = float(truncation)
        self.w5 = L.host_floats([weights["fs"], weights["center"], weights["tail"], weights["color"], weights["depth"]])
        self.mode = {"original": 0, "no_mask": 1}[mask_mode]
        self.perturb = perturb
        self.group = group
        self.bwd_mode = bwd_mode
        self.bound = bound.to(dev)
        self.bhost = bound_host(bound)
        self.t_uni = torch.linspace(0., 1., steps=n_stratified).to(dev)
        self.t_surf = torch.linspace(0., 1., steps=n_importance).to(dev)
        self.lr = dict(lr)
        self.probe = None               # dict name -> [(start_event, end_event)]: per-kernel HIP-event timing (bench.py)
        self.probe_every = 1            # with self.probe set: every k-th iteration is a probing one (events around the
        self._it, self._probing = 0, False   # launches, both branches on ONE stream so that durations are the kernels' own)
        self._adopt_parameters()
        self._alloc(max_rays)
        self.reset_optimizer(1.0)

    # ------------------------------------------------------------------------------------------ parameters
    def _adopt_parameters(self):
        dec, dev = self.dec, self.device
        self.desc_s, self.desc_c = dec.mlp_descs()
        n_s, n_c = mlp_n_params(self.desc_s), mlp_n_params(self.desc_c)
        self.has_beta = isinstance(dec.beta, nn.Parameter)
        self.o_dec_s, self.o_dec_c, self.o_beta = 0, n_s, n_s + n_c
        self.n_dec = _align(n_s + n_c + 1)
        self.o_tab_s = self.n_dec
        self.o_tab_c = self.o_tab_s + _align(self.es.desc.n_params)
        self.n_flat = self.o_tab_c + _align(self.ec.desc.n_params)
        flat = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
        grad = torch.zeros_like(flat)

        def adopt(p, off, shape=None):
            n = p.numel()
            view = flat[off:off + n].view(p.shape if shape is None else shape)
            view.copy_(p.detach())
            p.data = view
            p.grad = grad[off:off + n].view(view.shape)

        if dec.tcnn_network:
            adopt(dec.sdf_decoder.params, self.o_dec_s)
            adopt(dec.color_decoder.params, self.o_dec_c)
        else:
            for base, hidden, out in ((self.o_dec_s, dec.linears, dec.output_linear),
                                      (self.o_dec_c, dec.c_linears, dec.c_output_linear)):
                o = base
                for l in hidden:
                    adopt(l.weight, o); o += l.weight.numel()
                adopt(out.weight, o); o += 16 * out.weight.shape[1]          # rows n_out..15 stay zero (padding)
                for l in hidden:
                    adopt(l.bias, o); o += l.bias.numel()
                adopt(out.bias, o); o += 16
        if self.has_beta:
            adopt(dec.beta, self.o_beta)
        else:
            flat[self.o_beta] = float(dec.beta)
        adopt(self.es.params, self.o_tab_s)
        adopt(self.ec.params, self.o_tab_c)
        self.flat, self.grad = flat, grad
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.step_dev = torch.zeros(8, dtype=torch.float32, device=flat.device)    # Adam's step count, kept on the device (graph replay)
        self._graph = None

    def reset_optimizer(self, lr_factor=1.0):
        """Mapper.py:358-364: a fresh Adam for every mapped frame (moments and step count restart)."""
        self._join_side_streams()
        self.m.zero_(); self.v.zero_()
        self.opt_step = 0
        self.step_dev.zero_()
        # step_dev[1]: the optimiser's EPOCH (how many times it was reset), read by the arena window's in-kernel pixel draw and jitter: a graph
        # that is replayed for every mapped frame carries one host-side seed, and the step count restarts with every frame
        self._epoch = getattr(self, "_epoch", 0) + 1
        self.step_dev[1] = float(self._epoch)
        self._step_advanced = False
        self.lr_factor = float(lr_factor)
        self._graph = None              # a captured iteration holds the old learning rates

    def _join_side_streams(self):
        """the main stream waits for whatever a previous call left on the scan / side streams (the scans of a forward pass whose backward
        pass never came read the workspace; a queued step increment touches step_dev)"""
        # (only streams that hold work since their last join: a wait for an idle side stream is harmless when launched eagerly, but inside
        #  a hipGraph capture it would tie the capturing stream to an event recorded outside the capture)
        if getattr(self, "_scan_pending", False) and self.scan_stream is not None:
            torch.cuda.current_stream().wait_stream(self.scan_stream)
        if getattr(self, "_side_pending", False) and self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self._scan_pending = self._side_pending = False

    # ------------------------------------------------------------------------------------------ buffers
    def _alloc(self, R):
        # generation: bumped whenever buffers a captured graph may hold change their addresses (MapWindow compares it before a replay)
        self.generation = getattr(self, "generation", 0) + 1
        self._join_side_streams()
        self._step_advanced = False
        dev, S = self.device, self.S
        N = R * S
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.max_rays = R
        self.dydx_s = self.dydx_c = self.dpts_s = self.dpts_c = None
        self._graph = None              # a captured iteration holds the old buffers' addresses: capture() again after a reallocation
        self.z, self.pts = f(R, S), f(R, S, 3)
        self.feat_s, self.feat_c = f(N, 32), f(N, 32)
        self.raw, self.d_raw = f(R, S, 4), f(R, S, 4)
        self.d_feat_s, self.d_feat_c = f(N, 32), f(N, 32)
        self.term, self.unc, self.depth, self.dunc, self.rgb = f(R), f(R), f(R), f(R), f(R, 3)
        self.g_sdf, self.g_depth, self.g_rgb = f(R, S), f(R), f(R, 3)
        self.partials = f(int(L.lib().us_loss_partials_size(R)))
        self.stats, self.loss = f(10), f(1)
        self.beta_part = f(R)
        self.valid = torch.empty(R, dtype=torch.uint8, device=dev)
        lib = L.lib()
        da, db = ctypes.byref(self.es.desc), ctypes.byref(self.ec.desc)
        one_grid = lambda n: max(int(lib.us_hashgrid_bwd_workspace_bytes(da, n)), int(lib.us_hashgrid_bwd_workspace_bytes(db, n)))
        joint_ok = lambda n: bool(self._joint_wanted and self.bwd_mode in (-1, 3) and lib.us_hashgrid_joint_supported(da, db, n))
        binned_ok = lambda n: bool(lib.us_hashgrid_bwd_binned_supported(da, n) and lib.us_hashgrid_bwd_binned_supported(db, n))
        # scratch of the path actually taken: ONE set for the joint kernels, two (one per branch: they run on two streams) for the
        # one-grid kernels; max_workspace_bytes bounds a set.  A batch whose set would be larger (or whose records outgrow the 32-bit
        # record addresses) is walked in ranges of chunk_rays rays; a TABLE beyond the bin budget takes the LDS-sliced kernels.
        need = lambda n: int(lib.us_hashgrid_joint_workspace_bytes(da, db, n)) if joint_ok(n) else one_grid(n)
        self.chunk_rays = 0                                # > 0: the table gradient walks the batch in ranges of this many rays
        if self.bwd_mode in (-1, 3):
            parts = 1
            while True:
                r_k = -(-R // parts)
                if ((joint_ok(r_k * S) or binned_ok(r_k * S)) and need(r_k * S) <= self.max_ws) or r_k <= 16:
                    break
                parts += 1
            if parts > 1:
                self.chunk_rays = r_k
                N = r_k * S                                # the scratch below is sized for one range
            if self.bwd_mode == -1 and not (joint_ok(N) or binned_ok(N)):
                self.bwd_mode, self.chunk_rays, N = 1, 0, R * S
        self.joint = joint_ok(N)
        self.ws_bytes = need(N) if self.bwd_mode in (-1, 3) else 0
        mk_ws = lambda: torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev) if self.bwd_mode in (-1, 3) else None
        self.ws = mk_ws()
        self.ws_s = self.ws if (self.joint or self.ws is None) else mk_ws()    # the forward pass leaves each branch's binning counts in its own
        self.mlp_ws_bytes = max(int(lib.us_mlp_bwd_workspace_bytes(ctypes.byref(self.desc_s))),
                                int(lib.us_mlp_bwd_workspace_bytes(ctypes.byref(self.desc_c))))
        self.mlp_ws = torch.empty(self.mlp_ws_bytes, dtype=torch.uint8, device=dev)
        self.mlp_ws_s = torch.empty(self.mlp_ws_bytes, dtype=torch.uint8, device=dev)      # (one per decoder: they may run in one launch)

    def _split_flags(self, joint_branch):
        """(encoder flag, decoder flag) of the pre-split feature hand-over (r6): in the joint branches, with two split-bf16 decoders of 32
        inputs, the encoder writes the features as the decoders' hi / lo bf16 operand pairs (same bytes, same values: the decoders skip
        the split of their inputs, forward and backward).  Nothing else reads the feature planes.  feat_split = False turns it off."""
        on = bool(joint_branch and getattr(self, "feat_split", L.FEAT_SPLIT_DEFAULT) and self.desc_s.precision == 1 and self.desc_c.precision == 1
                  and self.desc_s.n_in == 32 and self.desc_c.n_in == 32)
        self._gs, self._ms = (L.US_GRID_FEAT_SPLIT_BF16, L.US_MLP_IN_SPLIT_BF16) if on else (0, 0)
        return self._gs, self._ms

    def _act_flags(self):
        """the output activations' hand-over (r6): with bf16-family decoders the decoder launches write / take PRE-activation values
        (US_MLP_OUT_PREACT / US_MLP_DOUT_PREACT) and the compositing launches apply tanh / sigmoid and their derivatives
        (US_RENDER_ACT in `mode`): ~80 VALU instructions per 32 points leave kernels that are bound by VALU issue for kernels that wait on
        their loads.  raw is rewritten in place with the activated samples by us_render_loss_fwd: same values as before.
        act_handover = False turns it off."""
        on = bool(getattr(self, "act_handover", L.ACT_HANDOVER_DEFAULT) and self.desc_s.precision != 0 and self.desc_c.precision != 0)
        if on:
            self._mo, self._md = L.US_MLP_OUT_PREACT, L.US_MLP_DOUT_PREACT
            self._ra = L.US_RENDER_ACT_ON | (int(self.desc_c.out_act) << 12) | (int(self.desc_s.out_act) << 16)
        else:
            self._mo = self._md = self._ra = 0
        return self._mo

    def _decoder_pair(self):
        """joint path: run the two decoders as ONE launch each way (us_mlp_fwd_pair / us_mlp_bwd_pair)?  Yes when they have one shape and a
        bf16 precision."""
        return bool(self.decoder_pair and
                    L.lib().us_mlp_pair_supported(ctypes.byref(self.desc_s), ctypes.byref(self.desc_c)))

    class _Branch:
        """`with step._branch() as st2:` -- the launches inside go to the side stream (st2 = its handle), which first waits
        for everything queued on the main stream; _join() makes the main stream wait for the side stream.  Without overlap
        the block simply runs on the main stream."""
        def __init__(self, step, after=None):
            self.step, self.after = step, after

        def __enter__(self):
            s = self.step
            self.on = s.overlap and not s._probing
            if not self.on:
                return L.stream()
            if s.side is None:
                s.side = torch.cuda.Stream(device=s.device)
            if self.after is not None:                   # behind a point of the main stream that lies BEFORE launches already queued there
                s.side.wait_event(self.after)
            else:
                s.side.wait_stream(torch.cuda.current_stream())
            s._side_pending = True
            self.ctx = torch.cuda.stream(s.side)
            self.ctx.__enter__()
            return L.stream()

        def __exit__(self, *a):
            if self.on:
                self.ctx.__exit__(*a)
            return False

    def _branch(self, after=None):
        return MapStep._Branch(self, after)

    def _join(self):
        if self.overlap and not self._probing and self.side is not None and self._side_pending:
            torch.cuda.current_stream().wait_stream(self.side)
            self._side_pending = False

    def _wait_scans(self):
        if self.scan_stream is not None and self._scan_pending:
            torch.cuda.current_stream().wait_stream(self.scan_stream)
            self._scan_pending = False

    def _timed(self, name, rc_fn):
        """run one C-ABI launch; with self.probe set, bracket it with HIP events on the launch stream"""
        if not self._probing:
            L.check(rc_fn(), name)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(rc_fn(), name)
        e1.record()
        self.probe.setdefault(name, []).append((e0, e1))

    # ------------------------------------------------------------------------------------------ the iteration
    def _seed(self, k):
        """seed of the k-th in-kernel random stream of the current call (rng_calls counts the calls)"""
        return (self.rng_seed + 0x9E3779B97F4A7C15 * (3 * self.rng_calls + k)) & (2 ** 64 - 1)

    def forward(self, rays_o, rays_d, gt_depth, gt_color, t_rand=None, has_zero_depth=None, zero_depth_draws=None, backward_follows=True,
                presampled=False):
        """
        Sample, encode, decode, composite and reduce the LOCAL loss sums and counts into self.stats[10].
        has_zero_depth: False -> the caller knows every ray carries a depth (e.g. noted per keyframe when its pool was cut) and the
        branch's launches are skipped; None / True -> run the importance-sampling branch of Renderer.py:104-130 for the rays with
        gt_depth == 0 (their number stays on the device: no host synchronisation either way).
        zero_depth_draws: (t_rand_uni [n0, n_strat], u [n0, n_imp]) for that branch's rays in row order, to replay a given random stream
        (tests); default: the in-kernel generator.
        backward_follows: False for a render-only call (the encoders then skip the bookkeeping they do for the table gradient).
        presampled: the pre-filter flags, z and the unit-cube points of these rays are already in self.valid / self.z / self.pts
        (window.MapWindow forms rays and samples in one launch).
        Side effects with backward_follows (joint kernels, side streams): the call is the first half of ONE optimiser step -- it queues the
        binning's scans (which in OVERWRITE mode clear the entries of hot bins in self.grad) and Adam's step increment on the scan
        stream and notes that in self._step_advanced; backward() joins that stream and adam_step() consumes the note.  Pair every such
        forward() with backward() and adam_step() (iterate() / MapWindow do); use backward_follows=False for a render-only call.
        """
        lib, st = L.lib(), L.stream()
        self._probing = self.probe is not None and (self._it % max(1, self.probe_every) == 0)
        self._it += 1
        self._dydx_valid = False
        o, d, gd, gc = L.f32(rays_o.detach()), L.f32(rays_d.detach()), L.f32(gt_depth.detach()), L.f32(gt_color.detach())
        R, S = o.shape[0], self.S
        if R > self.max_rays:
            self._alloc(R)
        N = R * S
        P = L.ptr
        off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
        mo_ = self._act_flags()
        # pre-filter against the scene box (Mapper.py:396-406) as a validity flag instead of a compaction
        c_free, s_off, s_span = ctypes.c_float(1.2), ctypes.c_float(1.5 * self.truncation), ctypes.c_float(3 * self.truncation)
        # filter + z + points in one launch; jitter from t_rand or, if none is given, from the in-kernel generator
        tr = P(L.f32(t_rand)) if (self.perturb and t_rand is not None) else None
        seed = self._seed
        if not presampled:                                       # presampled: the caller's kernel (us_window_sample) has filled valid, z and
            self.rng_calls += 1                                  # pts for these rays and advanced rng_calls
            L.check(lib.us_sample_points(P(o), P(d), P(gd), self.bhost, R, P(self.t_uni), self.n_strat, P(self.t_surf), self.n_imp,
                                         c_free, s_off, s_span, tr, seed(0), P(self.step_dev),
                                         1 if self.perturb else 0, 0, P(self.valid), P(self.z), P(self.pts), st), "us_sample_points")
        fl = self.flat
        if has_zero_depth is not False:
            # Renderer.py:104-130 for the rays without a depth measurement, on their compacted rows: coarse uniform pass through the
            # sdf grid + decoder, importance samples, and the rows of z / pts rewritten in place.  The row count stays on the device
            # (us_zero_depth_resample: every launch is sized for R rows and reads the count there), so the branch costs no host
            # synchronisation and a window with depth holes can be captured like any other.
            Su = self.n_strat
            if self.zd_rows is None or self.zd_rows.numel() < R:
                self.zd_rows = torch.empty(self.max_rays, dtype=torch.int32, device=self.device)
                self.zd_count = torch.zeros(1, dtype=torch.int32, device=self.device)
                self.zd_z = torch.empty(self.max_rays * Su, dtype=torch.float32, device=self.device)
                self.zd_sdf, self.zd_pts = torch.empty_like(self.zd_z), torch.empty(self.max_rays * Su * 3, dtype=torch.float32, device=self.device)
            tr0, u0 = zero_depth_draws if zero_depth_draws is not None else (None, None)
            tr0 = L.f32(tr0) if (tr0 is not None and self.perturb) else None
            u0 = L.f32(u0) if u0 is not None else None
            feat = self.d_feat_s                            # free until the backward pass; R * Su <= R * S rows
            L.check(lib.us_zero_depth_resample(ctypes.byref(self.es.desc), off(fl, self.o_tab_s), ctypes.byref(self.desc_s), off(fl, self.o_dec_s),
                                               off(fl, self.o_beta), P(o), P(d), P(gd), R, self.bhost, P(self.t_uni), Su, self.n_imp,
                                               P(tr0) if tr0 is not None else None, P(u0) if u0 is not None else None, seed(1), seed(2),
                                               P(self.step_dev), 1 if self.perturb else 0, P(self.zd_rows), P(self.zd_count), P(self.zd_z), P(self.zd_pts),
                                               P(feat), P(self.zd_sdf), P(self.z), P(self.pts), st), "us_zero_depth_resample")
        fl = self.flat
        ds, dc = ctypes.byref(self.es.desc), ctypes.byref(self.ec.desc)
        ms, mc = ctypes.byref(self.desc_s), ctypes.byref(self.desc_c)
        # the sdf and the colour branch are independent between the sample points and the compositing: two streams
        # with the binned backward the encoder also leaves the binning counts of these points in the branch's workspace
        # (us_hashgrid_fwd_counted: the gathers bound the kernel, the counting rides along), and the backward skips its count pass
        if self.chunk_rays:
            backward_follows = False                             # the counts of a forward pass belong to the whole batch, not to its ranges
        counted = self.ws is not None and self.count_in_forward and backward_follows
        self._counted = counted
        self._jcounted = False
        if self.joint and backward_follows:
            # both encoders in one launch (cells, positions and hashes computed once; the binning counts of both grids ride along),
            # then the two decoders side by side
            self._wait_scans()                                   # a scan of the previous call may still read the workspace
            self._jcounted = True
            self._dydx_valid = False
            gs_, ms_ = self._split_flags(True)
            if self.store_dydx:
                L_ = self.es.desc.n_levels
                if self.dydx_s is None or self.dydx_s.numel() < L_ * self.max_rays * S * 6:
                    self.dydx_s = torch.empty(L_ * self.max_rays * S * 6, dtype=torch.float16, device=self.device)
                    self.dydx_c = torch.empty_like(self.dydx_s)
                self._dydx_valid = True
                self._timed("hashgrid_fwd_joint", lambda: lib.us_hashgrid_fwd_joint_dydx(ds, dc, off(fl, self.o_tab_s), off(fl, self.o_tab_c), P(self.pts), N,
                                                                                         P(self.feat_s), P(self.feat_c), P(self.dydx_s), P(self.dydx_c), 3 | gs_,
                                                                                         P(self.ws), self.ws_bytes, st))
            else:
                self._timed("hashgrid_fwd_joint", lambda: lib.us_hashgrid_fwd_joint(ds, dc, off(fl, self.o_tab_s), off(fl, self.o_tab_c), P(self.pts), N,
                                                                                    P(self.feat_s), P(self.feat_c), 3 | gs_, P(self.ws), self.ws_bytes, st))
            # the binning's scan passes depend on the counts only: they run beside the decoders (own stream; the backward pass waits for
            # it), off the critical path.  A probed step keeps them on the one stream, timed by themselves.
            scan_call = lambda q: lib.us_hashgrid_joint_scan(ds, dc, N, off(self.grad, self.o_tab_s), off(self.grad, self.o_tab_c),
                                                             3 | L.US_GRID_BWD_OVERWRITE | self._det, P(self.ws), self.ws_bytes, q)
            # The critical chain is queued FIRST at a fork: in a captured graph the branch that is recorded first keeps the producer's
            # queue, and a kernel that changes queues starts ~10 us late (profiles/r03_timeline.txt).  So the decoders are launched
            # before the scans, which wait for an event recorded right behind the encoder.
            main_first = self._decoder_pair() and self.overlap and not self._probing
            ev = None
            if main_first:
                ev = torch.cuda.Event(); ev.record()

            def fork_scans():
                if self.scan_stream is None:
                    self.scan_stream = torch.cuda.Stream(device=self.device)
                if ev is not None:
                    self.scan_stream.wait_event(ev)
                else:
                    self.scan_stream.wait_stream(torch.cuda.current_stream())
                self._scan_pending = True
                with torch.cuda.stream(self.scan_stream):
                    # (the increment first: a one-thread launch behind the scans would be issued while the decoders' backward pass fills
                    #  every CU's LDS, and sit there until one of its workgroups leaves -- 47 us, profiles/r04_timeline.txt's forerunner)
                    if not self._step_advanced:                  # Adam's step count for this iteration (the sampler has read the old one)
                        L.check(lib.us_adam_step_inc(P(self.step_dev), 0.9, 0.999, L.stream()), "us_adam_step_inc")
                        self._step_advanced = True
                    L.check(scan_call(L.stream()), "us_hashgrid_joint_scan")
            if self._probing or not self.overlap:
                self._timed("hashgrid_scan_joint", lambda: scan_call(st))
            elif not main_first:
                fork_scans()
            if self._decoder_pair():                             # both decoders in one launch
                self._timed("mlp_fwd_pair", lambda: lib.us_mlp_fwd_pair(ms, mc, off(fl, self.o_dec_s), off(fl, self.o_dec_c), P(self.feat_s), P(self.feat_c), N,
                                                                        off(self.raw, 3), 4, P(self.raw), 4, 1 | ms_ | mo_, st))
                if main_first:
                    fork_scans()
                return self._finish_forward(o, d, gd, gc, R)
            # decoders of different shapes: one after the other on the main stream
            self._timed("mlp_fwd_sdf", lambda: lib.us_mlp_fwd(ms, off(fl, self.o_dec_s), P(self.feat_s), N, off(self.raw, 3), 4, 1 | ms_ | mo_, st))
            self._timed("mlp_fwd_color", lambda: lib.us_mlp_fwd(mc, off(fl, self.o_dec_c), P(self.feat_c), N, P(self.raw), 4, 1 | ms_ | mo_, st))
            return self._finish_forward(o, d, gd, gc, R)
        if not backward_follows and self.joint and self._decoder_pair():
            # a render-only call on ONE stream: both encoders in one launch (no binning counts), both decoders in one launch
            gs_, ms_ = self._split_flags(True)
            self._timed("hashgrid_fwd_joint", lambda: lib.us_hashgrid_fwd_joint(ds, dc, off(fl, self.o_tab_s), off(fl, self.o_tab_c), P(self.pts), N,
                                                                                P(self.feat_s), P(self.feat_c), 3 | gs_, None, 0, st))
            self._timed("mlp_fwd_pair", lambda: lib.us_mlp_fwd_pair(ms, mc, off(fl, self.o_dec_s), off(fl, self.o_dec_c), P(self.feat_s), P(self.feat_c), N,
                                                                    off(self.raw, 3), 4, P(self.raw), 4, 1 | ms_ | mo_, st))
            return self._finish_forward(o, d, gd, gc, R)
        self._split_flags(False)                                 # (the one-grid encoders write float planes)
        with self._branch() as st2:
            if counted:
                self._timed("hashgrid_fwd_sdf", lambda: lib.us_hashgrid_fwd_counted(ds, off(fl, self.o_tab_s), P(self.pts), N, P(self.feat_s), 3,
                                                                                    P(self.ws_s), self.ws_bytes, st2))
            else:
                self._timed("hashgrid_fwd_sdf", lambda: lib.us_hashgrid_fwd(ds, off(fl, self.o_tab_s), P(self.pts), N, P(self.feat_s), None, 3, st2))
            self._timed("mlp_fwd_sdf", lambda: lib.us_mlp_fwd(ms, off(fl, self.o_dec_s), P(self.feat_s), N, off(self.raw, 3), 4, 1 | mo_, st2))
        if counted:
            self._timed("hashgrid_fwd_color", lambda: lib.us_hashgrid_fwd_counted(dc, off(fl, self.o_tab_c), P(self.pts), N, P(self.feat_c), 3,
                                                                                  P(self.ws), self.ws_bytes, st))
        else:
            self._timed("hashgrid_fwd_color", lambda: lib.us_hashgrid_fwd(dc, off(fl, self.o_tab_c), P(self.pts), N, P(self.feat_c), None, 3, st))
        self._timed("mlp_fwd_color", lambda: lib.us_mlp_fwd(mc, off(fl, self.o_dec_c), P(self.feat_c), N, P(self.raw), 4, 1 | mo_, st))
        self._join()
        return self._finish_forward(o, d, gd, gc, R)

    def _finish_forward(self, o, d, gd, gc, R):
        lib, st, P, S, fl = L.lib(), L.stream(), L.ptr, self.S, self.flat
        beta = ctypes.c_void_p(fl.data_ptr() + 4 * self.o_beta)
        # compositing + the loss's sums and counts in one launch (+ the fixed-order reduction)
        L.check(lib.us_render_loss_fwd(P(self.raw), P(self.z), beta, R, S, self.mode | getattr(self, "_ra", 0), P(self.valid), P(gd), P(gc), self.truncation,
                                       P(self.term), P(self.unc), P(self.depth), P(self.rgb), P(self.dunc), P(self.partials), P(self.stats), st),
                "us_render_loss_fwd")
        self._batch = (o, d, gd, gc, R)
        self.n_rays = R
        return self.stats

    def backward(self, on_ready=None, ray_grads=False, fold=False):
        """
        Gradients of loss = sum_k w_k * sums_k / counts_k (self.stats, possibly reduced over ranks) into self.grad.
        ray_grads: also form dL/d(rays_o), dL/d(rays_d) (self.g_o, self.g_d [R,3]) -- what the joint pose optimisation of
        src/Mapper.py:358-374 differentiates through; the positions' gradient re-gathers the tables (no stored dy/dx).
        The colour branch runs first; on_ready(view) is called when the colour-table segment, and at the end the
        [decoders | beta | sdf table] segment, of self.grad are final (dist.dp_iterate overlaps their all-reduces).
        fold: the caller runs adam_step() next and reads no decoder gradient in between (iterate(), MapWindow): the decoder-gradient
        and beta reductions are then left to adam_step(), which sums them inside the decoders' optimiser launch
        (us_mlp_reduce_pair_adam) -- no side-stream work around the table gradient, no queue crossing in front of the record pass.
        """
        lib, st = L.lib(), L.stream()
        o, d, gd, gc, R = self._batch
        S, N = self.S, R * self.S
        P = L.ptr
        off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
        fl = self.flat
        ds, dc = ctypes.byref(self.es.desc), ctypes.byref(self.ec.desc)
        ms, mc = ctypes.byref(self.desc_s), ctypes.byref(self.desc_c)
        beta = off(fl, self.o_beta)
        # The binned table backward writes every table entry (US_GRID_BWD_OVERWRITE); the decoder segment, which the MLP
        # backward adds to, was cleared by the previous adam_step (zero_grad_mask) -- or is cleared here.
        binned = self.ws is not None
        self._folded = False
        self._tables_stepped = False
        self._grad_image_written = False                         # only the branch that writes the bf16 image in THIS call sets it (dist.GradComm.announce)
        # Single process, joint grids, two streams: the small reductions of the backward pass (decoder-gradient partials, d(beta), Adam's
        # step count) are taken off the critical path -- they run on the side stream beside the table gradient instead of ahead of it.
        defer = bool(self.joint and binned and not self.chunk_rays and self.overlap and not self._probing)
        clear_later = False
        if self.bwd_mode not in (-1, 3):
            self.grad.zero_()
        elif not self._dec_grad_clean:
            if defer:
                clear_later = True                               # ... and so is the clearing of the segment they add to (a 6 us fill)
            else:
                self.grad[:self.o_tab_s].zero_()
        self._dec_grad_clean = False
        gbeta = off(self.grad, self.o_beta) if self.has_beta else None
        # the loss gradients (from the possibly all-reduced statistics) + the compositing backward in one launch
        md_ = getattr(self, "_md", 0)                            # (the forward pass's choice: dL/d(raw) leaves w.r.t. the pre-activation outputs)
        L.check(lib.us_render_loss_bwd(P(self.raw), P(self.z), beta, R, S, self.mode | getattr(self, "_ra", 0) | (L.US_LOSS_DEFER_BETA if (defer and gbeta is not None) else 0),
                                       P(self.valid), P(gd), P(gc), P(self.depth), P(self.rgb),
                                       P(self.unc), self.truncation, self.w5, P(self.stats), P(self.d_raw), gbeta, P(self.beta_part),
                                       P(self.loss), st), "us_render_loss_bwd")

        def sdf_branch(q):
            self._timed("mlp_bwd_sdf", lambda: lib.us_mlp_bwd(ms, off(fl, self.o_dec_s), P(self.feat_s), off(self.raw, 3), 4,
                                                              off(self.d_raw, 3), 4, N, P(self.d_feat_s), off(self.grad, self.o_dec_s), 1 | md_,
                                                              P(self.mlp_ws_s), self.mlp_ws_bytes, q))
            if binned:
                self._timed("hashgrid_bwd_sdf", lambda: lib.us_hashgrid_bwd_binned(ds, P(self.pts), P(self.d_feat_s), N, off(self.grad, self.o_tab_s),
                                                                                  3 | L.US_GRID_BWD_OVERWRITE | self._det | (L.US_GRID_BWD_COUNTED if self._counted else 0), P(self.ws_s), self.ws_bytes, q))
            else:
                self._timed("hashgrid_bwd_sdf", lambda: lib.us_hashgrid_bwd_params(ds, P(self.pts), P(self.d_feat_s), N,
                                                                                  off(self.grad, self.o_tab_s), self.bwd_mode, 3, q))

        def color_branch(q):
            self._timed("mlp_bwd_color", lambda: lib.us_mlp_bwd(mc, off(fl, self.o_dec_c), P(self.feat_c), P(self.raw), 4, P(self.d_raw), 4,
                                                                N, P(self.d_feat_c), off(self.grad, self.o_dec_c), 1 | md_, P(self.mlp_ws), self.mlp_ws_bytes, q))
            if binned:
                self._timed("hashgrid_bwd_color", lambda: lib.us_hashgrid_bwd_binned(dc, P(self.pts), P(self.d_feat_c), N, off(self.grad, self.o_tab_c),
                                                                                    3 | L.US_GRID_BWD_OVERWRITE | self._det | (L.US_GRID_BWD_COUNTED if self._counted else 0), P(self.ws), self.ws_bytes, q))
            else:
                self._timed("hashgrid_bwd_color", lambda: lib.us_hashgrid_bwd_params(dc, P(self.pts), P(self.d_feat_c), N,
                                                                                    off(self.grad, self.o_tab_c), self.bwd_mode, 3, q))

        if self.chunk_rays and binned:
            self._backward_in_ranges(R, on_ready)
        elif self.joint:
            # the two decoder backward passes side by side, then ONE binned pass for both tables
            mflags = 1 | md_ | getattr(self, "_ms", 0) | (L.US_MLP_DEFER_REDUCE if defer else 0)   # (_ms: the forward pass left pre-split feature planes)
            mlp_s = lambda q: self._timed("mlp_bwd_sdf", lambda: lib.us_mlp_bwd(ms, off(fl, self.o_dec_s), P(self.feat_s), off(self.raw, 3), 4,
                                                                                off(self.d_raw, 3), 4, N, P(self.d_feat_s), off(self.grad, self.o_dec_s), mflags,
                                                                                P(self.mlp_ws_s), self.mlp_ws_bytes, q))
            self._dpts_valid = False
            if self._decoder_pair() and ray_grads and self._dydx_valid:
                # ... which also contract dL/d(features) with the encoder's dy/dx while it is in registers: each decoder's share of
                # dL/d(point) (the pose gradient's input: no second pass over dL/d(features))
                if getattr(self, "dpts_s", None) is None or self.dpts_s.numel() < self.max_rays * S * 3:
                    self.dpts_s = torch.empty(self.max_rays * S * 3, dtype=torch.float32, device=self.device)
                    self.dpts_c = torch.empty_like(self.dpts_s)
                self._dpts_valid = True
                self._timed("mlp_bwd_pair", lambda: lib.us_mlp_bwd_pair_dydx(
                    ms, mc, off(fl, self.o_dec_s), off(fl, self.o_dec_c), P(self.feat_s), P(self.feat_c), off(self.raw, 3), 4, P(self.raw), 4,
                    off(self.d_raw, 3), 4, P(self.d_raw), 4, N, P(self.d_feat_s), P(self.d_feat_c), off(self.grad, self.o_dec_s),
                    off(self.grad, self.o_dec_c), mflags, P(self.mlp_ws_s), P(self.mlp_ws), self.mlp_ws_bytes, P(self.dydx_s), P(self.dydx_c),
                    P(self.dpts_s), P(self.dpts_c), st))
            elif self._decoder_pair():                           # both decoders' backward passes in one launch
                self._timed("mlp_bwd_pair", lambda: lib.us_mlp_bwd_pair(ms, mc, off(fl, self.o_dec_s), off(fl, self.o_dec_c), P(self.feat_s), P(self.feat_c),
                                                                        off(self.raw, 3), 4, P(self.raw), 4, off(self.d_raw, 3), 4, P(self.d_raw), 4, N,
                                                                        P(self.d_feat_s), P(self.d_feat_c), off(self.grad, self.o_dec_s),
                                                                        off(self.grad, self.o_dec_c), mflags, P(self.mlp_ws_s), P(self.mlp_ws),
                                                                        self.mlp_ws_bytes, st))
            else:
                mlp_s(st)
                self._timed("mlp_bwd_color", lambda: lib.us_mlp_bwd(mc, off(fl, self.o_dec_c), P(self.feat_c), P(self.raw), 4, P(self.d_raw), 4,
                                                                    N, P(self.d_feat_c), off(self.grad, self.o_dec_c), mflags, P(self.mlp_ws), self.mlp_ws_bytes, st))
            self._folded = bool(fold and defer and on_ready is None and self.group is None and self._decoder_pair())
            if self._folded:
                defer_side = False                               # adam_step() sums the partial rows itself
            else:
                defer_side = defer
            # (at THIS fork the side work is recorded first: queueing the record pass ahead of the reductions, as forward() does with the
            #  decoders, made the replayed graph serialise scans and reductions on one queue in front of the record pass: 0.596 ms against 0.546)
            def side_reductions(st2):                            # the decoder gradients' and beta's sums (+ the step count, if still to come)
                if clear_later:                                  # the decoder gradients' segment: first touched by the reductions below
                    self.grad[:self.o_tab_s].zero_()
                if self._decoder_pair():                         # (the pair launch's partial rows: reduced by its own function)
                    L.check(lib.us_mlp_reduce_pair(ms, mc, P(self.mlp_ws_s), P(self.mlp_ws), self.mlp_ws_bytes, N, off(self.grad, self.o_dec_s),
                                                   off(self.grad, self.o_dec_c), st2), "us_mlp_reduce_pair")
                else:
                    L.check(lib.us_mlp_reduce(ms, P(self.mlp_ws_s), self.mlp_ws_bytes, N, off(self.grad, self.o_dec_s), st2), "us_mlp_reduce")
                    L.check(lib.us_mlp_reduce(mc, P(self.mlp_ws), self.mlp_ws_bytes, N, off(self.grad, self.o_dec_c), st2), "us_mlp_reduce")
                if gbeta is not None:
                    L.check(lib.us_beta_reduce(P(self.beta_part), R, gbeta, st2), "us_beta_reduce")
                if not self._step_advanced:
                    L.check(lib.us_adam_step_inc(P(self.step_dev), 0.9, 0.999, st2), "us_adam_step_inc")
                    self._step_advanced = True
            # The data-parallel step (on_ready) keeps them on the MAIN stream, between the two accumulate launches: there they run under the
            # colour segment's all-reduce, which is the long pole anyway, and the record pass follows the decoders on their queue without a
            # fork (the side branch took the producer's queue in the captured graph and the record pass started 12 us late on the other).
            inline_side = defer_side and on_ready is not None
            if defer_side and not inline_side:
                with self._branch() as st2:                      # side stream, behind both decoders: beside the table gradient
                    side_reductions(st2)
            self._wait_scans()
            jflags = 3 | L.US_GRID_BWD_OVERWRITE | self._det | ((L.US_GRID_BWD_COUNTED | L.US_GRID_BWD_SCANNED) if self._jcounted else 0)
            if on_ready is not None:
                # someone waits for the segments (the data-parallel step): the record pass for both grids, then the accumulate pass per
                # grid, colour first -- its 44.7 MB all-reduce starts while the sdf table is still being summed
                self._grad_image_written = False
                if self._grad_image and (jflags & L.US_GRID_BWD_DETERMINISTIC):
                    # ... and the colour table's gradient comes out of the sweep as the bfloat16 image the all-reduce carries (no narrowing pass)
                    img = ctypes.c_void_p(self._grad_bf16.data_ptr() + 2 * self.o_tab_c)
                    self._timed("hashgrid_bwd_joint", lambda: lib.us_hashgrid_bwd_joint_img(
                        ds, dc, P(self.pts), P(self.d_feat_s), P(self.d_feat_c), N, off(self.grad, self.o_tab_s), off(self.grad, self.o_tab_c), img,
                        jflags | L.US_GRID_BWD_ONLY_B, P(self.ws), self.ws_bytes, st))
                    self._grad_image_written = True
                else:
                    self._timed("hashgrid_bwd_joint", lambda: lib.us_hashgrid_bwd_joint(ds, dc, P(self.pts), P(self.d_feat_s), P(self.d_feat_c), N,
                                                                                        off(self.grad, self.o_tab_s), off(self.grad, self.o_tab_c),
                                                                                        jflags | L.US_GRID_BWD_ONLY_B, P(self.ws), self.ws_bytes, st))
                on_ready(self.grad[self.o_tab_c:])
                if inline_side:
                    side_reductions(st)
                self._timed("hashgrid_bwd_joint_sdf", lambda: lib.us_hashgrid_bwd_joint(
                    ds, dc, P(self.pts), P(self.d_feat_s), P(self.d_feat_c), N, off(self.grad, self.o_tab_s), off(self.grad, self.o_tab_c),
                    jflags | L.US_GRID_BWD_ONLY_A | L.US_GRID_BWD_RECORDS_READY, P(self.ws), self.ws_bytes, st))
            elif fold and self.fuse_adam and self.group is None:
                # the tables' optimiser step rides in the accumulate pass's sweep; adam_step() then only takes the decoders' (and poses') group
                if not self._step_advanced:                      # (one-stream mode: the forward pass queued no increment)
                    L.check(lib.us_adam_step_inc(P(self.step_dev), 0.9, 0.999, st), "us_adam_step_inc")
                    self._step_advanced = True
                self._tables_stepped = True
                f = self.lr_factor
                ta = L.TableAdamDesc(off(fl, self.o_tab_s), off(self.m, self.o_tab_s), off(self.v, self.o_tab_s), off(fl, self.o_tab_c),
                                     off(self.m, self.o_tab_c), off(self.v, self.o_tab_c), self.lr["sdf_grid"] * f, self.lr["color_grid"] * f, 0.9, 0.999, 1e-8,
                                     P(self.step_dev), 1 if self.keep_table_grad else 0)
                self._timed("hashgrid_bwd_joint", lambda: lib.us_hashgrid_bwd_joint_adam(ds, dc, P(self.pts), P(self.d_feat_s), P(self.d_feat_c), N,
                                                                                         off(self.grad, self.o_tab_s), off(self.grad, self.o_tab_c),
                                                                                         ctypes.byref(ta), jflags, P(self.ws), self.ws_bytes, st))
            else:
                self._timed("hashgrid_bwd_joint", lambda: lib.us_hashgrid_bwd_joint(ds, dc, P(self.pts), P(self.d_feat_s), P(self.d_feat_c), N,
                                                                                    off(self.grad, self.o_tab_s), off(self.grad, self.o_tab_c), jflags,
                                                                                    P(self.ws), self.ws_bytes, st))
            if defer_side and not inline_side:
                self._join()                                     # ... and the deferred reductions are in before anything reads the gradients
        elif self.overlap and not self._probing:
            with self._branch() as st2:                          # sdf branch on the side stream, colour branch beside it
                sdf_branch(st2)
            color_branch(st)
            if on_ready is not None:
                on_ready(self.grad[self.o_tab_c:])
            self._join()
        else:                                                    # one stream: colour first, so that its (large) gradient
            color_branch(st)                                     # segment can travel while the sdf branch computes
            if on_ready is not None:
                on_ready(self.grad[self.o_tab_c:])
            sdf_branch(st)
        if on_ready is not None:
            on_ready(self.grad[:self.o_tab_c])
        if ray_grads:
            self._ray_gradients(R)
        self.n_rays = R
        return self.loss

    def _ray_gradients(self, R):
        """dL/d(rays_o), dL/d(rays_d) of the last backward pass into self.g_o / self.g_d (on the current stream)"""
        lib, st, P, S, N, fl = L.lib(), L.stream(), L.ptr, self.S, R * self.S, self.flat
        off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
        ds, dc = ctypes.byref(self.es.desc), ctypes.byref(self.ec.desc)
        if not hasattr(self, "g_o") or self.g_o.shape[0] < R:
            f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=self.device)
            self.g_o, self.g_d, self.d_pts = f(self.max_rays, 3), f(self.max_rays, 3), None
        if getattr(self, "_dpts_valid", False):
            # the decoders' backward launch has left the two grids' shares of dL/d(point): add, reduce to the rays
            self._timed("ray_points_bwd2", lambda: lib.us_ray_points_bwd2(P(self.dpts_s), P(self.dpts_c), P(self.z), self.bhost, R, S, P(self.g_o),
                                                                          P(self.g_d), st))
        elif self._dydx_valid and S <= 128:
            # the forward pass left dy/dx: one streaming launch contracts it with dL/dy and reduces to the rays
            self._timed("hashgrid_dydx_rays", lambda: lib.us_hashgrid_dydx_rays(self.es.desc.n_levels, P(self.d_feat_s), P(self.d_feat_c), P(self.dydx_s),
                                                                                P(self.dydx_c), R, S, P(self.z), self.bhost, P(self.g_o), P(self.g_d),
                                                                                None, st))
        elif getattr(self, "_tables_stepped", False):
            raise L.UniSlamHipError("MapStep: ray gradients after fuse_adam need the encoder's dy/dx (store_dydx, S <= 128): the gathering fallbacks would "
                                    "read tables the accumulate pass's sweep has already stepped")
        elif lib.us_hashgrid_bwd_input_rays_supported(ds, dc, S):
            # both grids' input gradient and its reduction to the rays in ONE launch (no [N,3] round trip, no second gather launch)
            self._timed("hashgrid_bwd_input_rays", lambda: lib.us_hashgrid_bwd_input_rays(
                ds, dc, off(fl, self.o_tab_s), off(fl, self.o_tab_c), P(self.pts), P(self.d_feat_s), P(self.d_feat_c), R, S, P(self.z),
                self.bhost, P(self.g_o), P(self.g_d), None, 3, st))
        else:
            if self.d_pts is None or self.d_pts.shape[0] < R:
                self.d_pts = torch.empty((self.max_rays, S, 3), dtype=torch.float32, device=self.device)
            L.check(lib.us_hashgrid_bwd_input_gather(ds, off(fl, self.o_tab_s), P(self.pts), P(self.d_feat_s), N, P(self.d_pts), 3, st),
                    "us_hashgrid_bwd_input_gather")
            L.check(lib.us_hashgrid_bwd_input_gather(dc, off(fl, self.o_tab_c), P(self.pts), P(self.d_feat_c), N, P(self.d_pts),
                                                     3 | L.US_GRID_ACCUMULATE, st), "us_hashgrid_bwd_input_gather")
            L.check(lib.us_ray_points_bwd(P(self.d_pts), P(self.z), self.bhost, R, S, P(self.g_o), P(self.g_d), st), "us_ray_points_bwd")

    def _backward_in_ranges(self, R, on_ready):
        """the decoders' backward passes over the whole batch, then the table gradients range by range (scratch within max_workspace_bytes)"""
        lib, st, P, S, N, fl = L.lib(), L.stream(), L.ptr, self.S, R * self.S, self.flat
        off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
        ds, dc = ctypes.byref(self.es.desc), ctypes.byref(self.ec.desc)
        ms, mc = ctypes.byref(self.desc_s), ctypes.byref(self.desc_c)
        mf = 1 | getattr(self, "_ms", 0) | getattr(self, "_md", 0)
        L.check(lib.us_mlp_bwd(mc, off(fl, self.o_dec_c), P(self.feat_c), P(self.raw), 4, P(self.d_raw), 4, N, P(self.d_feat_c),
                               off(self.grad, self.o_dec_c), mf, P(self.mlp_ws), self.mlp_ws_bytes, st), "us_mlp_bwd")
        L.check(lib.us_mlp_bwd(ms, off(fl, self.o_dec_s), P(self.feat_s), off(self.raw, 3), 4, off(self.d_raw, 3), 4, N, P(self.d_feat_s),
                               off(self.grad, self.o_dec_s), mf, P(self.mlp_ws_s), self.mlp_ws_bytes, st), "us_mlp_bwd")
        for k, r0 in enumerate(range(0, R, self.chunk_rays)):
            n_k, i0 = (min(R, r0 + self.chunk_rays) - r0) * S, r0 * S
            flags = 3 | self._det | (L.US_GRID_BWD_OVERWRITE if k == 0 else 0)
            x, dya, dyb = off(self.pts, 3 * i0), off(self.d_feat_s, 2 * i0), off(self.d_feat_c, 2 * i0)
            if self.joint:
                L.check(lib.us_hashgrid_bwd_joint_range(ds, dc, x, dya, dyb, n_k, N, off(self.grad, self.o_tab_s), off(self.grad, self.o_tab_c),
                                                        flags, P(self.ws), self.ws_bytes, st), "us_hashgrid_bwd_joint_range")
            else:
                L.check(lib.us_hashgrid_bwd_binned_range(dc, x, dyb, n_k, N, off(self.grad, self.o_tab_c), flags, P(self.ws),
                                                         self.ws_bytes, st), "us_hashgrid_bwd_binned_range")
                L.check(lib.us_hashgrid_bwd_binned_range(ds, x, dya, n_k, N, off(self.grad, self.o_tab_s), flags, P(self.ws_s),
                                                         self.ws_bytes, st), "us_hashgrid_bwd_binned_range")
        if on_ready is not None:
            on_ready(self.grad[self.o_tab_c:])

    def forward_backward(self, rays_o, rays_d, gt_depth, gt_color, t_rand=None, has_zero_depth=None, ray_grads=False, zero_depth_draws=None):
        """single-process forward + backward (no optimiser step); returns loss[1]"""
        self.forward(rays_o, rays_d, gt_depth, gt_color, t_rand, has_zero_depth, zero_depth_draws)
        return self.backward(ray_grads=ray_grads)

    def ray_gradients(self):
        """(dL/d rays_o [R,3], dL/d rays_d [R,3]) of the last backward(ray_grads=True); rays dropped by the pre-filter get 0"""
        R = self.n_rays
        return self.g_o[:R], self.g_d[:R]

    def enable_grad_image(self, on):
        """dist.GradComm (payload "bf16_colour", this engine's optimiser reads the image itself): have the accumulate pass of the colour table
        leave its gradient as the bfloat16 image (self._grad_bf16) as well.  That launch writes every entry exactly once: no bin is split
        over workgroups (US_GRID_BWD_DETERMINISTIC from here on, in the scan passes too).  Returns whether the path is taken."""
        ok = bool(on) and self.joint and self.dp_mode == "local_fast"
        if ok and not self._grad_image:
            self._det_before_image = self._det_base               # (what the constructor asked for: not the fuse_adam bit of the property)
            self._det = L.US_GRID_BWD_DETERMINISTIC
        elif not ok and self._grad_image:
            self._det = getattr(self, "_det_before_image", 0)    # what the constructor's `deterministic` asked for
        self._grad_image = ok
        return ok

    def adam_step(self, ranges=None, part=None, poses=None):
        """
        torch.optim.Adam over the three param groups (Mapper.py:118-126) in one launch.  ranges: None (everything) or a list of
        (lo, hi) index ranges of the flat buffer -- the shards this rank owns when the optimiser state is sharded over ranks.
        part: None | "colour" | "rest" -- ONE optimiser step as two launches (the data-parallel step: the colour table's pass, 47 of the
        54 us, runs as soon as ITS gradient segment has arrived and covers the reduction of the small remaining segment; "colour" first).
        """
        lib, st, P = L.lib(), L.stream(), L.ptr
        if part != "rest":
            self.opt_step += 1
        f = self.lr_factor
        groups = ((0, self.n_dec, self.lr["decoders"] * f), (self.o_tab_s, self.es.desc.n_params, self.lr["sdf_grid"] * f),
                  (self.o_tab_c, self.ec.desc.n_params, self.lr["color_grid"] * f))
        stepped, self._tables_stepped = self._tables_stepped, False     # backward(fold=True) with fuse_adam: the tables' step is done
        if stepped and (ranges is not None or part is not None):
            raise L.UniSlamHipError("MapStep.adam_step: the tables were stepped inside the accumulate pass (fuse_adam); ranges / parts do not apply")
        if stepped:
            groups = (groups[0], (self.o_tab_s, 0, 0.0), (self.o_tab_c, 0, 0.0))
        if ranges is None and getattr(self, "_folded", False):
            # backward(fold=True) left the decoders' partial rows and beta's per-ray partials: their sums and the decoder group's Adam
            # in one launch, then the tables
            self._folded = False
            if not self._step_advanced:                          # (a forward pass that did not queue the increment: one-stream mode)
                L.check(lib.us_adam_step_inc(P(self.step_dev), 0.9, 0.999, st), "us_adam_step_inc")
                self._step_advanced = True
            off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
            hb = self.has_beta
            dec_args = (ctypes.byref(self.desc_s), ctypes.byref(self.desc_c), P(self.mlp_ws_s), P(self.mlp_ws), self.mlp_ws_bytes, self.n_rays * self.S,
                        off(self.flat, self.o_dec_s), off(self.flat, self.o_dec_c), off(self.grad, self.o_dec_s), off(self.grad, self.o_dec_c),
                        off(self.m, self.o_dec_s), off(self.m, self.o_dec_c), off(self.v, self.o_dec_s), off(self.v, self.o_dec_c),
                        P(self.beta_part) if hb else None, self.n_rays, off(self.flat, self.o_beta) if hb else None,
                        off(self.grad, self.o_beta) if hb else None, off(self.m, self.o_beta) if hb else None, off(self.v, self.o_beta) if hb else None,
                        self.lr["decoders"] * f)
            self._dec_grad_clean = False
            if self.one_launch_adam:
                # the whole optimiser step in ONE launch: the decoder group's reductions + Adam ride as three slices of the tables' launch
                tabs = list(groups)[1:]
                if stepped:                                      # the accumulate pass's sweep has applied the tables' step (fuse_adam)
                    tabs = [(self.o_tab_s, 0, 0.0), (self.o_tab_c, 0, 0.0)]
                I64, DBL = ctypes.c_int64 * 2, ctypes.c_double * 2
                L.check(lib.us_adam_step_model(*dec_args, P(self.flat), P(self.grad), P(self.m), P(self.v), 2, I64(*[g[0] for g in tabs]),
                                               I64(*[g[1] for g in tabs]), DBL(*[g[2] for g in tabs]), 0.9, 0.999, 1e-8, P(self.step_dev),
                                               L.US_ADAM_STEP_ADVANCED, ctypes.byref(poses) if poses is not None else None, st), "us_adam_step_model")
                self._step_advanced = False
                return
            L.check(lib.us_mlp_reduce_pair_adam(*dec_args, 0.9, 0.999, 1e-8, P(self.step_dev), st), "us_mlp_reduce_pair_adam")
            segs, zero_mask = list(groups)[1:], 0
        elif ranges is None and part == "colour":
            segs, zero_mask = [groups[2]], 0
        elif ranges is None and part == "rest":
            segs, zero_mask = list(groups[:2]), 0b001 | L.US_ADAM_STEP_ADVANCED      # (the colour part advanced the step count)
            self._dec_grad_clean = True
        elif ranges is None:
            segs, zero_mask = list(groups), 0b001                # the decoder gradients (which the MLP backward adds to) are
            self._dec_grad_clean = True                          # cleared on the way
        else:
            segs, zero_mask = [], 0
            for (lo, hi) in ranges:
                for (o, n, lr) in groups:
                    a, b = max(lo, o), min(hi, o + n)
                    if b > a:
                        segs.append((a, b - a, lr))
            self._dec_grad_clean = False
        if not segs:
            segs = [(0, 0, 0.0)]        # a rank that owns only padding still advances the device-side step count (k_step_inc), so the
        k = len(segs)                   # bias corrections and the sampler's jitter salt stay in lock-step over the ranks
        if self._step_advanced:         # backward() already queued the step increment (us_adam_step_inc, beside the table gradient)
            zero_mask |= L.US_ADAM_STEP_ADVANCED
            self._step_advanced = False
        I64, DBL = ctypes.c_int64 * k, ctypes.c_double * k
        # the step count lives on the device (advanced by the launch itself): nothing in the arguments changes between iterations
        narrow = getattr(self, "_grad_bf16_from", None)          # dist.GradComm: segments from this index on arrived as bfloat16 (engine._grad_bf16)
        if narrow is not None and ranges is None:
            mask = sum(1 << i for i, g in enumerate(segs) if g[0] >= narrow)
            L.check(lib.us_adam_step_segments_bf16(P(self.flat), P(self.grad), P(self._grad_bf16), mask, P(self.m), P(self.v), k,
                                                   I64(*[g[0] for g in segs]), I64(*[g[1] for g in segs]), DBL(*[g[2] for g in segs]), 0.9, 0.999,
                                                   1e-8, P(self.step_dev), zero_mask, st), "us_adam_step_segments_bf16")
            return
        L.check(lib.us_adam_step_segments_dev(P(self.flat), P(self.grad), P(self.m), P(self.v), k, I64(*[g[0] for g in segs]),
                                              I64(*[g[1] for g in segs]), DBL(*[g[2] for g in segs]), 0.9, 0.999, 1e-8, P(self.step_dev),
                                              zero_mask, st), "us_adam_step_segments_dev")

    def iterate(self, rays_o, rays_d, gt_depth, gt_color, t_rand=None, has_zero_depth=None, presampled=False, zero_depth_draws=None):
        """One full mapping iteration (Mapper.py:366-445 minus ray selection). Returns the loss as a device tensor [1]."""
        if self.group is None:
            self.forward(rays_o, rays_d, gt_depth, gt_color, t_rand, has_zero_depth, zero_depth_draws, True, presampled)
            loss = self.backward(fold=True)
            self.adam_step()
            return loss
        return dp_iterate(self, (rays_o, rays_d, gt_depth, gt_color, t_rand, has_zero_depth, zero_depth_draws, True, presampled), self.group)

    def capture(self, n_rays, t_rand=False):
        """
        Capture one iteration (rays without the zero-depth branch, single process) into a hipGraph; replay() then runs it with one
        host call, and the two branch streams are scheduled by the graph instead of by events (0.706 -> 0.685 ms at 4096 x 64).
        Returns the static input tensors (rays_o [n,3], rays_d [n,3], gt_depth [n], gt_color [n,3][, t_rand [n,S]]): write the
        next batch INTO them (e.g. let common.get_samples_all's kernel target them), then call replay().  The jitter comes from the
        in-kernel generator (varied per replay by the device-side step count) unless t_rand=True.  Adam's step count is on the
        device, so a replay advances the optimiser exactly as an eager iterate() does; eager and replayed iterations can be mixed.
        """
        from .graph import CapturedIteration
        if self.group is not None:
            raise L.UniSlamHipError("MapStep.capture: single-process only (the data-parallel step waits on RCCL work handles)")
        if self._step_advanced:
            raise L.UniSlamHipError("MapStep.capture: a forward pass is pending (its step increment is queued): finish the optimiser step "
                                    "(backward + adam_step) first -- the captured iteration carries its own increment")
        self._join_side_streams()
        dev = self.device
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        ins = [f(n_rays, 3), f(n_rays, 3) + 1.0, f(n_rays) + 1.0, f(n_rays, 3)]
        tr = f(n_rays, self.S) if t_rand else None
        was, self.probe = self.probe, None
        # warm-up iterations would move the parameters: run them with the learning rates at zero and restore the state after
        keep = (self.flat.clone(), self.m.clone(), self.v.clone(), self.step_dev.clone(), self.opt_step, dict(self.lr), self.rng_calls)
        self.lr = {k: 0.0 for k in self.lr}
        fn = lambda: self.iterate(ins[0], ins[1], ins[2], ins[3], t_rand=tr, has_zero_depth=False)

        def restore():                  # everything the warm-up iterations and the traced call touched -- also when one of them raises
            self.lr = keep[5]
            self.flat.copy_(keep[0]); self.m.copy_(keep[1]); self.v.copy_(keep[2]); self.step_dev.copy_(keep[3])
            self.opt_step, self.rng_calls = keep[4], keep[6]
            self._dec_grad_clean = False    # the captured backward clears the decoder gradient itself: a replay is then valid after any call
            self._step_advanced, self._folded = False, False

        graph, s = None, None
        try:
            try:
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(2):
                        fn()
            finally:
                if s is not None:
                    torch.cuda.current_stream().wait_stream(s)
                restore()               # the real learning rates are what the capture records
            graph = CapturedIteration(fn, warmup=0)
        finally:
            restore()                   # (the capture pass does not execute; its host-side counters are undone)
            self.probe = was
        self._graph = graph
        return tuple(ins) + ((tr,) if t_rand else ())

    def replay(self):
        """run the captured iteration on what the static input tensors hold now; returns the loss tensor [1] (static)"""
        if self._graph is None:
            raise L.UniSlamHipError("MapStep.replay: call capture() first (and again after reset_optimizer())")
        self.opt_step += 1
        return self._graph.replay()

    def rendered(self):
        """views of the last iteration's per-ray outputs: (term, pixel_unc, depth, rgb, sdf, z_vals, depth_unc)"""
        R = self.n_rays
        return (self.term[:R], self.unc[:R], self.depth[:R], self.rgb[:R], self.raw[:R, :, 3], self.z[:R], self.dunc[:R])
