#!/usr/bin/env python3
"""
bench.py -- throughput of Uni-SLAM's mapping iteration (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or as the plain command above, in which case this process starts the N rank processes
itself (one per GPU, RCCL), relays rank 0's JSON line and exits non-zero if any rank did.  The launching process never touches the GPU.

One "step" = one mapping iteration of the hot path (src/Mapper.py:366-445): draw a fresh batch of pixels from the keyframe pools
(common.get_samples_all -> us_gather_rays, src/Mapper.py:379-393) -> bbox pre-filter -> z sampling (+jitter) -> points -> 2x
hash-grid encode -> 2x fused MLP -> SDF->alpha compositing -> uncertainty-gated loss -> full backward (table, decoder and beta
gradients) -> [all-reduce over ranks] -> Adam over all 12.9 M parameters.  Workload (config.workload): BASELINE configs[1] =
Replica room0 geometry, 4096 rays x 64 samples (48 stratified + 16 surface), L=16 F=2 hash grids (log2T 16 sdf / 19 colour,
finest resolution 816), 2 hidden x 32 MLP decoders; the batch is 256 pixels from each of 16 synthetic keyframes (10 % pixel pools of
680x1200 frames, resident in HBM).  Weak scaling: every rank renders its own 4096 rays from its own keyframes (configs[3]).

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant kernel, timed with HIP events in a separate pass
after the timed region) and `cpu_baseline` (the CPU oracle port timed on the host cores, rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROOM0_BOUND = [[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]]           # configs/Replica/room0.yaml:3
W = dict(fs=5, center=200, tail=10, color=5, depth=0.1)          # configs/UNISLAM.yaml:67-71
LR = dict(decoders=0.001, sdf_grid=0.05, color_grid=0.05)        # configs/Replica/replica.yaml:19-21
HBM_PEAK_GBS = 8000.0                                            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
CAM = dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5)   # configs/Replica/replica.yaml:36-41
N_KEYFRAMES = 16                                                 # 16 x 256 pixels = 4096 rays (Mapper.py:315: pixels // frames)
POOL_FRACTION = 0.1                                              # Mapper.py:  10 % of a keyframe's pixels are kept as its pool

torch = None                                                     # imported by the rank processes only (run_rank)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--hidden", type=int, default=32, help="MLP width (32 = BASELINE '2x32'; 16 = reference decoders.py default)")
    ap.add_argument("--bwd-mode", type=int, default=-1)
    ap.add_argument("--mlp-precision", default="bf16", choices=["fp32", "bf16", "bf16_plain", "f16"],
                    help="MFMA operand type of the decoders.  bf16 (headline): v_mfma_f32_16x16x32_bf16 with split operands (hi + lo) in the "
                         "forward products -- rendered depth / colour within 4e-5 of the fp32 decoders on identical parameters "
                         "(tools/bf16_deviation.py; bound 1e-3), bf16 operands in the gradient products; fp32: f32-input MFMA throughout")
    ap.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16", "bf16_colour"],
                    help="payload type of the gradient all-reduce (N > 1).  fp32 (default): N ranks reproduce one process on the concatenated batch "
                         "to rounding -- the reference's fp32 Adam input; bf16_colour (lossy, opt-in): the colour table's segment -- 44.7 of the 51.7 MB "
                         "-- travels as bfloat16, the geometry (sdf table, decoders, beta) as fp32: a 15-iteration window converges to the same map "
                         "(held-out depth 7e-7, colour 8e-6 from fp32 gradients; tests/test_gpu_step.py); bf16: everything narrow")
    ap.add_argument("--sharded-adam", action="store_true",
                    help="N > 1: reduce-scatter the gradient, Adam on this rank's shard, all-gather the parameters")
    ap.add_argument("--dp-mode", default="local_fast", choices=["local_fast", "colour_first"],
                    help="N > 1: local_fast = the single-process kernels with the accumulate pass split per grid (0.59 ms of rank-local work, the "
                         "colour table's all-reduce hides behind ~55 us); colour_first = one-grid kernels, colour branch before sdf branch (0.67 ms, "
                         "~176 us of cover): the better choice once that all-reduce takes more than ~130 us (DESIGN.md 7)")
    ap.add_argument("--no-overlap", action="store_true", help="run the sdf and colour branches on one stream")
    ap.add_argument("--no-decoder-pair", action="store_true", help="the two decoders as two launches each way instead of one (MapStep.decoder_pair)")
    ap.add_argument("--joint", default="auto", choices=["auto", "0", "1"],
                    help="both grids in one encoder launch and one binned table-gradient pass (csrc/hashgrid_joint.hip); auto = MapStep's default")
    ap.add_argument("--no-graph", action="store_true", help="launch every iteration eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--dp-graph", default="between", choices=["inside", "between"],
                    help="N > 1: hipGraph segments with the collectives eager between them (graph.SegmentedGraph; default -- the form the world-2 "
                         "tests cover), or the step as ONE hipGraph with the RCCL calls captured inside it (0.52 instead of 0.62 ms of rank-local "
                         "work, but only ever run on a 1-rank group: opt in on a node where it can be watched)")
    ap.add_argument("--fixed-batch", action="store_true", help="re-render one fixed batch every step (round 1's bench) instead of a fresh draw")
    ap.add_argument("--probe-steps", type=int, default=10, help="eager iterations with HIP events around the hot kernels, after the timed region")
    ap.add_argument("--prewarm-s", type=float, default=0.3, help="seconds of untimed iterations before the W warm-up steps (clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-tracking", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (forward only, bf16 decoders, trained-like tables)")
    ap.add_argument("--side", default="", choices=["", "dp_rank_local"],
                    help="(internal) run ONE side measurement and print its JSON: what side_process() starts as a child of the N = 1 run")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N > 1 from a plain command line
def spawn_ranks(args, argv):
    """
    `python bench.py --gpus N` without torchrun: start N copies of this script, one per GPU, with the torchrun environment
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT).  This process makes no GPU call (it does not even import
    torch); it relays rank 0's stdout (the JSON line) and returns the first non-zero exit code of any rank.
    """
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = subprocess.PIPE if r == 0 else sys.stderr          # only rank 0 prints the record
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out, stderr=sys.stderr))
    rc, line0 = 0, b""
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)     # rank 0's stdout is drained while all ranks are watched
    reader.start()
    try:
        live = list(procs)
        while live and rc == 0:                                    # the first non-zero exit ends the job: a rank that outlives a failed
            for p in list(live):                                   # peer would wait in a collective (or in the rendezvous) until its timeout
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    if code != 0:
                        rc = code
            if live and rc == 0:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    reader.join(timeout=10)
    line0 = out0[0] if out0 else b""
    for line in line0.decode(errors="replace").splitlines():       # stdout carries the JSON record only; library chatter
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")   # (e.g. gloo's connection notes) goes to stderr
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------ synthetic scene and inputs
def load_bound(bound, dividable=0.24):
    """src/UNISLAM.py:205-218"""
    b = torch.tensor(bound, dtype=torch.float64).float()
    b[:, 1] = (((b[:, 1] - b[:, 0]) / dividable).int() + 1) * dividable + b[:, 0]
    return b


def per_level_scale(res, n_levels=16):
    return float(2.0 ** (math.log2(res / n_levels) / (n_levels - 1)))      # src/UNISLAM.py:241


def _rotation(q):
    r, i, j, k = q.tolist()
    return torch.tensor([[1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r)],
                         [2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r)],
                         [2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)]])


def synthetic_rays(R, bound, seed, device):
    """one fixed batch: a camera at the scene centre with a seeded random rotation, Replica intrinsics, U(0.5,3.5) m depths"""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(4, generator=g); q = q / q.norm()
    Rm = _rotation(q)
    u = torch.randint(CAM["W"], (R,), generator=g).float(); v = torch.randint(CAM["H"], (R,), generator=g).float()
    dirs = torch.stack([(u - CAM["cx"]) / CAM["fx"], -(v - CAM["cy"]) / CAM["fy"], -torch.ones(R)], -1)
    rays_d = dirs @ Rm.t()
    centre = bound.mean(dim=1)
    rays_o = centre.expand(R, 3).contiguous()
    t = (bound.unsqueeze(0) - rays_o.unsqueeze(-1)) / rays_d.unsqueeze(-1)
    far = torch.min(torch.max(t, dim=2)[0], dim=1)[0]
    depth = torch.minimum(torch.rand(R, generator=g) * 3.0 + 0.5, 0.9 * far)
    color = torch.rand(R, 3, generator=g)
    return rays_o.to(device), rays_d.to(device), depth.to(device), color.to(device)


def keyframe_pools(n_frames, bound, seed, device):
    """
    What Mapper.optimize_mapping holds per selected keyframe (src/Mapper.py:315-356): a camera-to-world pose and a pool of
    POOL_FRACTION of the frame's pixels with their camera-frame directions, depths and colours, all resident on the device.
    Cameras sit around the scene centre with seeded random rotations; depths are U(0.5, 3.5) m clipped inside the scene box.
    Returns c2ws [b,4,4], depths [b,P], colors [b,P,3], dirs [b,P,3].
    """
    g = torch.Generator().manual_seed(seed)
    P = int(CAM["H"] * CAM["W"] * POOL_FRACTION)
    centre, half = bound.mean(dim=1), (bound[:, 1] - bound[:, 0]) / 2
    c2ws, depths, colors, dirs_all = [], [], [], []
    for _ in range(n_frames):
        q = torch.randn(4, generator=g); q = q / q.norm()
        Rm = _rotation(q)
        pos = centre + (torch.rand(3, generator=g) - 0.5) * half * 0.5
        c2w = torch.eye(4); c2w[:3, :3] = Rm; c2w[:3, 3] = pos
        pix = torch.randperm(CAM["H"] * CAM["W"], generator=g)[:P]
        u, v = (pix % CAM["W"]).float(), (pix // CAM["W"]).float()
        dirs = torch.stack([(u - CAM["cx"]) / CAM["fx"], -(v - CAM["cy"]) / CAM["fy"], -torch.ones(P)], -1)
        rd = dirs @ Rm.t()
        t = (bound.unsqueeze(0) - pos.reshape(1, 3, 1)) / rd.unsqueeze(-1)
        far = torch.min(torch.max(t, dim=2)[0], dim=1)[0]
        depths.append(torch.minimum(torch.rand(P, generator=g) * 3.0 + 0.5, 0.9 * far))
        colors.append(torch.rand(P, 3, generator=g)); dirs_all.append(dirs); c2ws.append(c2w)
    st = lambda xs: torch.stack(xs).contiguous().to(device)
    return st(c2ws), st(depths), st(colors), st(dirs_all)


def step_stats(fn, steps, warmup):
    """side-run timing, the headline's way: `warmup` untimed calls, `steps` calls between two synchronisations (wall time per step), then
    `steps` more with one HIP event pair per call on the launch stream -> (ms per step, last return value, {median, p10, p90} of the events)"""
    import gc
    out = None
    for _ in range(warmup):
        out = fn()
    # Python's cyclic collector is held off inside the timed loops: a full collection in a process that holds torch and a few models costs
    # 50 - 94 ms (measured: tools/time_dropin.py, `slowest_host_call_ms` / `max_ms` of single runs) and lands in one of every ~150 eager
    # autograd iterations -- inside a 100-step loop or not, which made the eager side runs jump by a factor two from run to run
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        return _step_stats_loops(fn, steps, out)
    finally:
        if was_enabled:
            gc.enable()


def _step_stats_loops(fn, steps, out):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t_prev, host_max = t0, 0.0
    for _ in range(steps):
        out = fn()
        t_now = time.perf_counter(); host_max = max(host_max, t_now - t_prev); t_prev = t_now
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a_, b_ in evs:
        a_.record(); out = fn(); b_.record()
    torch.cuda.synchronize()
    ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
    q = lambda f: ts[min(len(ts) - 1, int(f * len(ts)))]
    return ms, out, {"n": steps, "median_ms": q(0.5), "p10_ms": q(0.1), "p90_ms": q(0.9), "max_ms": ts[-1], "slowest_host_call_ms": 1e3 * host_max,
                     "python_gc": "held off inside the timed loops"}


def gather_rate_curve(path=None):
    """[(table bytes, G lane-loads/s)] of RANDOM 8-byte gathers by table size, as tools/ta_bench.hip measured them on the MI355X
    (profiles/r06_ta_bench.txt: one 1024-thread workgroup per CU, every lane its own address; 16-byte gathers run at the same rates)"""
    import re
    path = path or os.path.join(ROOT, "profiles", "r06_ta_bench.txt")
    pts = []
    if os.path.exists(path):
        for ln in open(path):
            m = re.match(r"random\s+8-byte gathers, table\s+([0-9.]+) MiB:.*?([0-9.]+) G lane-loads/s", ln)
            if m:
                pts.append((float(m.group(1)) * 2 ** 20, float(m.group(2))))
    return sorted(pts)


def _rate_at(curve, nbytes):
    """log-log interpolation of the curve (flat beyond its ends)"""
    if nbytes <= curve[0][0]:
        return curve[0][1]
    for (b0, r0), (b1, r1) in zip(curve, curve[1:]):
        if nbytes <= b1:
            t = (math.log(nbytes) - math.log(b0)) / (math.log(b1) - math.log(b0))
            return math.exp(math.log(r0) + t * (math.log(r1) - math.log(r0)))
    return curve[-1][1]


def encoder_request_roofline(descs, pts, kernel_ms):
    """
    The encoder against the bound it actually sits at: the rate at which the memory system serves RANDOM gather requests from a
    table of a level's size -- not HBM bytes (the tables live in the L2s / Infinity Cache; traffic is 0.6 x algorithmic).  Lane loads are
    counted from this batch's points exactly as gather_corners (csrc/hashgrid_dev.h) issues them: per point, level and grid 4 sixteen-byte
    gathers (the x / x+1 vertices of a cell edge together) -- plus 4 eight-byte ones on a HASHED level when the cell's x is odd (the pair is
    then not an aligned neighbour pair).  ceiling = sum over levels and grids of loads / rate(level's slab bytes), rate from the
    microbenchmark curve (gather_rate_curve: uniformly random addresses, every lane its own line); frac = ceiling time / measured time.
    A frac above 1 on coarse levels is expected (a ray's neighbouring samples share lines; the microbenchmark's lanes never do).
    """
    curve = gather_rate_curve()
    if not curve:
        return None
    x = pts.reshape(-1, 3).clamp(0, 1)
    n = x.shape[0]
    loads, t_model, by_level = 0, 0.0, []
    for d in descs:
        for l in range(d.n_levels):
            entries = int(d.offset[l + 1] - d.offset[l])
            res = int(d.resolution[l])
            hashed = res ** 3 > entries
            k = 4 * n
            if hashed:
                gx = torch.floor(x[:, 0] * float(d.scale[l]) + 0.5).to(torch.int64)
                k += 4 * int((gx & 1).sum())
            rate = _rate_at(curve, entries * 8)
            loads += k
            t_model += k / (rate * 1e9)
            by_level.append((l, entries * 8, k, rate))
    t = kernel_ms * 1e-3
    coarse = sum(k / (r * 1e9) for (_, b, k, r) in by_level if b <= 4 * 2 ** 20)
    return {"bound": "random gather requests (L2 / Infinity Cache request rate by table size)", "lane_loads_per_launch": loads,
            "achieved": loads / t / 1e9, "unit": "G lane-loads/s", "ceiling_ms": 1e3 * t_model, "frac": t_model / t,
            "ceiling_ms_levels_up_to_4MiB": 1e3 * coarse, "ceiling_ms_levels_beyond_4MiB": 1e3 * (t_model - coarse),
            "curve": "profiles/r06_ta_bench.txt (tools/ta_bench.hip; tools/gather_scope.hip: the same rate with loads that bypass the CU's cache)",
            "note": "ceiling = sum over levels and grids of lane loads / rate(slab bytes); uniformly random addresses in the microbenchmark, so "
                    "levels whose samples share lines run ahead of it"}


def _median(xs):
    """median launch duration of a probe pass (a first launch that pays for lazy set-up does not move it)"""
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def cpu_baseline(bound, n_strat, n_imp, hidden, rays, budget_s=20.0):
    """the CPU oracle port of the same iteration (oracle/unislam_oracle.py) on the host cores; bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import unislam_oracle as O
    # the GPU box gives one GPU a 16-core share of the host (oversubscribing its 256 hardware threads makes the CPU
    # run ~40x slower); use what the scheduler really grants, at most 16
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)           # read by libgomp when oracle/libhashgrid_ref.so is first used
    pls = per_level_scale(816)
    mk = lambda l2: O.HashGridOracle(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
                                         "log2_hashmap_size": l2, "base_resolution": 16, "per_level_scale": pls})
    es, ec = mk(16), mk(19)
    dec = O.DecodersOracle(c_dim=32, hidden_size=hidden, n_blocks=2)
    opt = torch.optim.Adam([{"params": list(dec.parameters()), "lr": LR["decoders"]},
                            {"params": [es.params], "lr": LR["sdf_grid"]}, {"params": [ec.params], "lr": LR["color_grid"]}])
    R = rays
    ro, rd, gd, gc = synthetic_rays(R, bound, 0, "cpu")
    it = lambda: O.mapping_iteration(([es], [ec]), dec, opt, ro, rd, gd, gc, bound, 0.06, n_strat, n_imp, W, "original", True)
    it()                                                    # warm-up
    t0 = time.perf_counter(); n = 0
    while True:
        it(); n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return {"value": R * n / el, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n} mapping iterations of {R} rays x {n_strat + n_imp} samples (same scene, tables and 2x{hidden} MLP; "
                      f"C hash grid with OpenMP + torch-CPU, {cores} threads), {el:.1f} s", "ms_per_iter": 1e3 * el / n}


def tracking_bench(us, es, ec, dec, bound, dev, iters=200):
    """
    secondary number (not the headline metric): one Tracker.optimize_tracking iteration (src/Tracker.py:149-244) at the
    Replica settings -- 2000 rays x 40 samples (configs/Replica/replica.yaml:12, UNISLAM.yaml:88-89), 680x1200 frame,
    ignore_edge 75, pose = Adam(lr_T 2e-3, lr_R 1e-3, betas (0.5, 0.999)) -- on the same room0 tables and decoders
    """
    H, Wd, fx, fy, cx, cy = CAM["H"], CAM["W"], CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"]
    g = torch.Generator().manual_seed(5)
    gt_depth = (torch.rand(1, H, Wd, generator=g) * 1.5 + 0.8).to(dev)
    gt_color = torch.rand(1, H, Wd, 3, generator=g).to(dev)
    centre = bound.mean(dim=1)
    pose = torch.tensor([[0.9, 0.1, -0.2, 0.3, float(centre[0]), float(centre[1]), float(centre[2])]], device=dev)
    ts = us.TrackStep(es, ec, dec, bound, 32, 8, 0.06, dict(fs=10, center=200, tail=50, color=5, depth=1), max_rays=2000)
    out = {"workload": "Replica tracking iteration (src/Tracker.py:149-244): 2000 rays x 40 samples of a 680x1200 frame, 10 x median gate, pose Adam; "
                       "nine HIP launches (TrackStep.iterate_fused: pixel draw + pose->rays + sampling | encoders + dy/dx | decoder pair | "
                       "compositing + loss partials | median gate + statistics | loss gradients + compositing backward | decoder pair backward | "
                       "dy/dx contraction to the rays | pose gradient + Adam)"}
    try:
        ts.begin_frame(pose[0], gt_color[0], gt_depth[0], 2e-3, 1e-3, H, Wd, fx, fy, cx, cy, 75, 75)
        fstep = lambda: ts.iterate_fused(2000)
        for _ in range(10):
            fstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss, _, _ = fstep()
        torch.cuda.synchronize()
        fms = 1e3 * (time.perf_counter() - t0) / iters
        out.update({"fused_eager_ms_per_iter": fms, "fused_eager_rays_per_s": 2000 / (fms / 1e3), "iters": iters, "final_loss": float(loss)})
        it = us.CapturedIteration(fstep)
        for _ in range(10):
            it.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss, _, _ = it.replay()
        torch.cuda.synchronize()
        fg = 1e3 * (time.perf_counter() - t0) / iters
        out.update({"fused_graph_ms_per_iter": fg, "fused_graph_rays_per_s": 2000 / (fg / 1e3), "fused_graph_final_loss": float(loss)})
        # the kernels' own durations (HIP events around every launch of 20 eager iterations) and the roofline entry of the dominant one
        ts.probe = {}
        for _ in range(20):
            fstep()
        torch.cuda.synchronize()
        kern = {k: _median([a.elapsed_time(b) for a, b in v]) for k, v in ts.probe.items()}
        ts.probe = None
        out["c_abi_calls_per_iteration"] = len(kern)
        out["launches_per_iteration"] = len(kern) + (1 if "us_track_loss_fwd" in kern else 0)     # us_track_loss_fwd is two kernels
        out["kernel_ms"] = {k: round(v, 4) for k, v in sorted(kern.items())}
        N = 2000 * 40
        enc = "us_hashgrid_fwd_joint_dydx"
        if enc in kern:
            alg = 2 * 1024 * N                                    # both encoders' gathers (1024 B per point and grid)
            ach = alg / (kern[enc] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": enc, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": kern[enc],
                               "note": "gather bytes only; the launch also writes 2 x 24 B per point and level of dy/dx for the pose gradient"}
    except Exception as e:                                # report, do not hide
        out["fused_error"] = repr(e)[:300]
    return out


def slam_bench(us, dev, hidden, prec, n_frames=30):
    """
    secondary number: what a FRAME costs through the drivers (unislam_amd.slam: Tracker.track_frame / Mapper.map_frame, after
    src/Tracker.py:271-370 and src/Mapper.py:461-545) at Replica's settings -- 680 x 1200 frames, 2000 rays x 8 tracking iterations per
    frame, 4000 rays x 15 mapping iterations every fourth frame over the keyframe window (joint_opt from the fifth keyframe), tables
    2^16 / 2^19 at 1 cm -- on the synthetic room (unislam_amd.synthetic; frames rendered ahead).  Wall time with a device
    synchronisation around every call; medians over the frames after the window has filled.
    """
    from unislam_amd.synthetic import SyntheticRoom
    from unislam_amd.slam import SLAM
    out = {"workload": f"{n_frames} frames 680x1200 of the synthetic room, Replica settings (tracking 2000 x 8, mapping 4000 x 15 every 4th frame, "
                       "joint_opt, tables 2^16 / 2^19 at 1 cm), drivers of unislam_amd.slam"}
    try:
        torch.manual_seed(0)
        frames = SyntheticRoom(n_frames=n_frames, H=CAM["H"], W=CAM["W"], device=dev)
        for i in range(n_frames):
            frames[i]
        bound = load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]]).to(dev)
        pls = per_level_scale(int((bound[:, 1] - bound[:, 0]).max() / 0.01))
        mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                                "base_resolution": 16, "per_level_scale": pls}).to(dev)
        cfg = {"rendering": {"perturb": True, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid",
               "grid": {"tcnn_network": False}, "model": {"mlp_precision": prec}}
        dec = us.Decoders(cfg, c_dim=32, hidden_size=hidden, truncation=0.06, n_blocks=2).to(dev)
        dec.bound = bound
        slam = SLAM(frames, (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy), mk(16), mk(19), dec, bound,
                    cfg={"mapping": dict(iters_first=100)})
        t_track, t_map = [], []

        def timed(fn, sink):
            def f(*a, **k):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = fn(*a, **k)
                torch.cuda.synchronize(); sink.append(1e3 * (time.perf_counter() - t0))
                return r
            return f

        slam.tracker.track_frame = timed(slam.tracker.track_frame, t_track)
        slam.mapper.map_frame = timed(slam.mapper.map_frame, t_map)
        slam.run()
        every = slam.cfg["mapping"]["every_frame"]
        tr, mp = _median(t_track[len(t_track) // 2:]), _median(t_map[len(t_map) // 2:])
        out.update({"tracking_ms_per_frame": tr, "mapping_ms_per_mapped_frame": mp, "mapped_every": every,
                    "steady_state_ms_per_frame": tr + mp / every, "steady_state_frames_per_s": 1e3 / (tr + mp / every),
                    "ate_rmse_cm": 100 * slam.ate_rmse(), "keyframes": len(slam.mapper.keyframe_list), "joint_opt": bool(slam.mapper.joint_opt),
                    "tracking_ms_all": [round(x, 2) for x in t_track], "mapping_ms_all": [round(x, 1) for x in t_map]})
        out["render_img"] = render_img_bench(us, slam, frames, cfg, bound, dev)
    except Exception as e:                                # report, do not hide
        out["error"] = repr(e)[:300]
    return out


def render_img_bench(us, slam, frames, cfg, bound, dev, reps=5):
    """
    the forward-only consumer of the path: Renderer.render_img (src/utils/Renderer.py:160-223) of one 680 x 1200 frame at
    its estimated pose on the map the run above built -- 816 000 rays x 40 samples, sampling + encoders + decoders + compositing per chunk,
    no gradient -- with the reference's chunk (ray_batch_size = 10000: 82 chunks) and with 204 000 rays per chunk (4 chunks).  Wall time
    around the call with a device synchronisation on both sides, median of `reps`; the PSNR of the render is a sanity number.
    """
    import types
    out = {}
    k = len(frames) - 1
    _, color, depth, _, _ = frames[k]
    c2w = slam.estimate_c2w_list[k].to(dev)
    holder = types.SimpleNamespace(bound=bound, device=dev, H=frames.H, W=frames.W, fx=frames.fx, fy=frames.fy, cx=frames.cx, cy=frames.cy)
    rcfg = dict(cfg, rendering=dict(cfg["rendering"], perturb=False))
    for name, chunk in (("ray_batch_size_10000", 10000), ("ray_batch_size_204000", 204000)):
        r = us.Renderer(rcfg, holder, ray_batch_size=chunk)
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d, c, *_ = r.render_img(([slam.es], [slam.ec]), slam.decoders, c2w, 0.06, dev, gt_depth=depth)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        ms = _median(ts[1:])
        mse = float(((c.float() - color) ** 2).mean())
        out[name] = {"ms_per_image": ms, "rays_per_s": frames.H * frames.W / (ms / 1e3), "chunks": -(-frames.H * frames.W // chunk),
                     "psnr_db": -10.0 * math.log10(max(mse, 1e-12))}
    out["workload"] = "Renderer.render_img of one 680 x 1200 frame (816 000 rays x 40 samples, forward only) at its estimated pose, bf16 decoders"
    return out


def joint_opt_bench(us, build_step, bound, dev, steps, warmup):
    """
    The reference's STEADY-STATE mapping iteration: joint_opt (configs/UNISLAM.yaml:50) is on from the fifth keyframe (src/Mapper.py:519),
    so the window's camera poses are a fourth Adam group and every iteration also runs poses -> rays and rays -> pose gradients
    (src/Mapper.py:359-376,443-459).  window.MapWindow: poses on the device, the iteration replayed from one hipGraph; a fresh pixel
    draw per step.  Two shapes: 4096 rays over 16 keyframes (the headline's batch), and the > 20-keyframe batch of src/Mapper.py:385-393
    (25 frames x 160 pixels + 10 x 200 extra rays from the newest frames = 6000 rays).
    """
    out = {}
    for tag, b, n_per, extra in (("4096_rays_16_keyframes", N_KEYFRAMES, 4096 // N_KEYFRAMES, None), ("6000_rays_25_keyframes_extra_10x200", 25, 160, (10, 200))):
        try:
            step = build_step()[0]
            c2ws, pd, pc, pr = keyframe_pools(b, bound, 2000, dev)
            win = us.MapWindow(step, c2ws, pd, pc, pr, n_per, joint_opt=True, cam_lr=1e-3, extra=extra, has_zero_depth=False)
            win.capture()
            for _ in range(warmup):
                win.replay()
            rounds = []
            for _ in range(3):                            # three timed rounds of `steps` replays, the median reported
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    loss = win.replay()
                torch.cuda.synchronize()
                rounds.append(1e3 * (time.perf_counter() - t0) / steps)
            ms = sorted(rounds)[1]
            ev = step_stats(win.replay, steps, 0)[2]
            moved = float((win.c2ws() - c2ws).abs().max())
            step.probe, step.probe_every, step._it = {}, 1, 0
            for _ in range(10):
                win.iterate()
            torch.cuda.synchronize()
            kern = {k: round(_median([a.elapsed_time(c) for a, c in v]), 4) for k, v in sorted(step.probe.items())}
            step.probe = None
            out[tag] = {"joint_opt_iteration_ms": ms, "rays": win.R, "rays_per_s": win.R / (ms / 1e3), "final_loss": float(loss),
                        "launch": "hipGraph replay of MapWindow (poses, pose Adam state and pixel indices on the device)",
                        "max_pose_matrix_change": moved, "kernel_ms": kern, "rounds_ms": [round(x, 4) for x in rounds], "step_time_hip_events": ev}
        except Exception as e:                            # report, do not hide
            out[tag] = {"error": repr(e)[:300]}
    return out


def dp_rank_local_bench(us, build_step, bound, dev, steps, warmup):
    """
    What ONE rank of the data-parallel step costs by itself: a 1-rank RCCL process group, so every collective of the step is
    issued and waited for but moves nothing.  MapStep(group=True) in its two dp_modes: eager, replayed as hipGraph segments between the
    collectives (graph.SegmentedGraph; the default), and replayed as ONE graph that holds the RCCL calls (opt-in); with the poses fixed and with
    joint_opt (rank-owned poses, src/Mapper.py:359-376).
    """
    import torch.distributed as dist
    out = {"note": "1-rank RCCL group on this GPU: collectives issued and waited for, no bytes moved; 4096 rays x 64 over 16 keyframes"}
    own_pg = not dist.is_initialized()
    try:
        if own_pg:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        c2ws, pd, pc, pr = keyframe_pools(N_KEYFRAMES, bound, 3000, dev)

        def timed(fn):
            for _ in range(warmup):
                fn()
            rounds = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    fn()
                torch.cuda.synchronize()
                rounds.append(1e3 * (time.perf_counter() - t0) / steps)
            return sorted(rounds)[1]

        for mode in ("local_fast", "colour_first"):
            o = {}
            for tag, jo in (("poses_fixed", False), ("joint_opt", True)):
                step = build_step(group=True, dp_mode=mode)[0]
                win = us.MapWindow(step, c2ws, pd, pc, pr, 4096 // N_KEYFRAMES, joint_opt=jo, cam_lr=1e-3, has_zero_depth=False)
                o[tag + "_eager_ms"] = timed(win.iterate)
                win.capture(collectives="between")
                o[tag + "_graph_segments_ms"] = timed(win.replay)
                o[tag + "_graph_segments"] = len(win._graph.segments)
                o[tag + "_graph_segments_launched"] = win._graph.n_launched    # (r6: a segment that recorded nothing is skipped on replay)
                win.capture(collectives="inside")            # RCCL: the collectives captured into the ONE graph (opt-in; this is why the run has a process of its own)
                o[tag + "_replayed_ms"] = timed(win.replay)
            out[mode] = o
        step = build_step(group=True, dp_mode="local_fast", sharded_adam=True)[0]
        win = us.MapWindow(step, c2ws, pd, pc, pr, 4096 // N_KEYFRAMES, joint_opt=False, has_zero_depth=False)
        win.capture(collectives="inside")
        out["local_fast_sharded_adam"] = {"poses_fixed_replayed_ms": timed(win.replay), "note": "reduce-scatter + Adam on this rank's shard "
                                          "(1 rank: the whole buffer) + all-gather"}
    except Exception as e:                                # report, do not hide
        out["error"] = repr(e)[:400]
    finally:
        if own_pg and dist.is_initialized():
            dist.destroy_process_group()
    return out


def side_process(name, args, timeout_s=300):
    """
    A side run in a process of its own, its JSON merged into the parent's record.  Used for `dp_rank_local_ms`: that run opens an RCCL
    process group and captures collectives into hipGraphs, and once (r6, 1 of ~40 runs) torch's process-group watchdog thread queried an event
    that had been recorded in a capturing stream (hipErrorCapturedEvent) and terminated the process -- which took the whole bench line with
    it.  A child that dies costs its own entry only: the error text is recorded and the run is tried once more.
    """
    cmd = [sys.executable, os.path.abspath(__file__), "--side", name, "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--rays", str(args.rays), "--hidden", str(args.hidden), "--mlp-precision", args.mlp_precision, "--bwd-mode", str(args.bwd_mode),
           "--joint", args.joint, "--grad-comm", args.grad_comm] + (["--no-overlap"] if args.no_overlap else []) + \
          (["--no-decoder-pair"] if args.no_decoder_pair else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    last = None
    for attempt in range(2):
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=os.path.dirname(os.path.abspath(__file__)))
        except subprocess.TimeoutExpired:
            return {"error": f"side process {name}: no result within {timeout_s} s"}
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            out = json.loads(lines[-1])
            if attempt:
                out["first_attempt"] = last
            return out
        err = [ln for ln in p.stderr.splitlines() if "rror" in ln]
        last = f"side process {name} ended with code {p.returncode}: " + (err[0] if err else p.stderr[-300:])[:400]
    return {"error": last}


def _tcnn_only_model_class():
    """(torch is imported by the rank processes only: the class is made on demand)"""
    class _TcnnOnlyModel(torch.nn.Module):
        """
        BENCH SCAFFOLDING for `drop_in_api.tcnn_only`: what a maintainer gets from the THREE changed lines of INTEGRATION.md 1 alone -- the
        reference's own torch code around `tcnn.Encoding` / `tcnn.Network`, restated here (the reference cannot travel to the GPU box):
        decoders as src/networks/decoders.py:49-70,91-105,118-128,143-153,182-205 (clamp -> enc(p) -> net(h) per decoder -> cat), rendering as
        src/utils/Renderer.py:42-57,81-101,132-152 (linspace, cat + sort, jitter, points, sdf2alpha, cat + cumprod, five reductions), sampling
        as src/common.py:152-166 (rotate ALL pool directions, then gather), pre-filter + four compactions + loss as src/Mapper.py:396-440 with
        sdf_losses :141-175.  Every op but the two tcnn modules is a torch op.
        """

        def __init__(self, tcnn, hidden, prec, log2T, res, bound, truncation, w, n_strat, n_imp):
            super().__init__()
            enc = lambda l2: tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
                                                                           "log2_hashmap_size": l2, "base_resolution": 16,
                                                                           "per_level_scale": per_level_scale(res)}, dtype=torch.float)
            net = lambda n_out, act: tcnn.Network(n_input_dims=32, n_output_dims=n_out, network_config={
                "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": hidden, "n_hidden_layers": 2,
                "precision": prec})                                   # ("precision": this build's extension key; 2 hidden x `hidden`: the bench's decoders)
            self.hash_grids_xyz, self.c_hash_grids_xyz = [enc(log2T[0])], [enc(log2T[1])]
            self.enc_s, self.enc_c = self.hash_grids_xyz[0], self.c_hash_grids_xyz[0]         # (registered; the reference keeps them in 1-element lists)
            self.sdf_decoder, self.color_decoder = net(1, "Tanh"), net(3, "Sigmoid")
            self.beta = torch.nn.Parameter(10 * torch.ones(1))
            self.bound, self.tr, self.w, self.n_strat, self.n_imp = bound, truncation, w, n_strat, n_imp

        def decoders(self, p):
            p_nor = p.reshape(-1, 3)
            sdf = self.sdf_decoder(self.hash_grids_xyz[0](torch.clamp(p_nor, min=0, max=1))).squeeze()
            rgb = self.color_decoder(self.c_hash_grids_xyz[0](torch.clamp(p_nor, min=0, max=1)))
            raw = torch.cat([rgb, sdf.unsqueeze(-1)], dim=-1)
            return raw.reshape(*p.shape[:-1], -1)

        def render(self, rays_d, rays_o, gt_depth):
            dev, tr = rays_o.device, self.tr
            t_uni = torch.linspace(0., 1., steps=self.n_strat, device=dev)
            t_surf = torch.linspace(0., 1., steps=self.n_imp, device=dev)
            gd = gt_depth.reshape(-1, 1)
            z_surf = gd.expand(-1, self.n_imp) - (1.5 * tr) + (3 * tr * t_surf)
            z_free = 0.0 + 1.2 * gd.expand(-1, self.n_strat) * t_uni
            z, _ = torch.sort(torch.cat([z_free, z_surf], dim=-1), dim=-1)
            mids = 0.5 * (z[..., 1:] + z[..., :-1])
            upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
            z = lower + (upper - lower) * torch.rand(z.shape, device=dev)
            pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
            pts = (pts - self.bound[:, 0]) / (self.bound[:, 1] - self.bound[:, 0])
            raw = self.decoders(pts)
            alpha = 1. - torch.exp(-self.beta * torch.sigmoid(-raw[..., 3] * self.beta))
            weights = alpha * torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1), device=dev), (1. - alpha + 1e-10)], -1), -1)[:, :-1]
            rgb = torch.sum(weights[..., None] * raw[..., :3], -2)
            depth = torch.sum(weights * z, -1)
            unc = torch.square(1 - torch.sum(weights, -1))
            return unc, depth, rgb, raw[..., 3], z

        def sdf_losses(self, sdf, z, gd):
            tr, w = self.tr, self.w
            front = torch.where(z < (gd[:, None] - tr), torch.ones_like(z), torch.zeros_like(z)).bool()
            back = torch.where(z > (gd[:, None] + tr), torch.ones_like(z), torch.zeros_like(z)).bool()
            center = torch.where((z > (gd[:, None] - 0.4 * tr)) * (z < (gd[:, None] + 0.4 * tr)), torch.ones_like(z), torch.zeros_like(z)).bool()
            tail = (~front) * (~back) * (~center)
            fs = torch.mean(torch.square(sdf[front] - torch.ones_like(sdf[front])))
            c = torch.mean(torch.square((z + sdf * tr)[center] - gd[:, None].expand(z.shape)[center]))
            t = torch.mean(torch.square((z + sdf * tr)[tail] - gd[:, None].expand(z.shape)[tail]))
            return w["fs"] * fs + w["center"] * c + w["tail"] * t

        def iteration(self, opt, c2ws, pd, pc, pr, n_per):
            dev = pd.device
            opt.zero_grad()
            b = c2ws.shape[0]
            idx = torch.randint(pd.shape[1], (n_per * b,), device=dev).reshape(b, -1)
            gd = torch.gather(pd, 1, idx)
            gc = torch.gather(pc, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
            rd = torch.sum(pr.unsqueeze(-2) * c2ws[:, None, :3, :3], -1)                      # all P directions of every frame, then the gather
            ro = c2ws[:, None, :3, -1].expand(rd.shape)
            rd = torch.gather(rd, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).reshape(-1, 3)
            ro = torch.gather(ro, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).reshape(-1, 3)
            gd, gc = gd.reshape(-1), gc.reshape(-1, 3)
            with torch.no_grad():
                t = (self.bound.unsqueeze(0) - ro.unsqueeze(-1)) / rd.unsqueeze(-1)
                t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
                inside = t >= gd
            rd, ro, gd, gc = rd[inside], ro[inside], gd[inside], gc[inside]
            unc, depth, color, sdf, z = self.render(rd, ro, gd)
            depth_mask = (gd > 0) & ((1 - unc.detach()) > 0.99)
            loss = self.sdf_losses(sdf[depth_mask], z[depth_mask], gd[depth_mask])
            loss = loss + self.w["color"] * torch.square(gc - color).mean()
            loss = loss + self.w["depth"] * torch.square(gd[depth_mask] - depth[depth_mask]).mean()
            loss.backward()
            opt.step()
            return loss.detach()
    return _TcnnOnlyModel


def drop_in_bench(us, dev, prec, hidden, bound, mk, steps, warmup, R, n_strat, n_imp):
    """
    BASELINE configs[1] through the REFERENCE'S OWN CALLS ONLY (what a maintainer gets from swapping the imports) --
    per iteration, eagerly, under torch autograd (src/Mapper.py:372-445):
        optimizer.zero_grad(); get_samples_all(...) -> bbox pre-filter -> Renderer.render_batch_ray(scene_rep, decoders, rays_d, rays_o, ...)
        -> mapping_loss(...) -> loss.backward() -> optimizer.step()
    with the reference's three param groups (src/Mapper.py:111-139).  Decoders.forward is one autograd node on the joint encoder, the
    decoder pair and the joint table gradient (decoders._DecodersFusedFn); the sampling is one launch.  Variants: the optimiser
    (torch.optim.Adam as the reference constructs it | unislam_amd.optim.Adam: one launch per step) and the pre-filter (the reference's
    boolean-mask compaction, a host synchronisation | a validity flag handed to the loss).
    """
    import types
    out = {"workload": f"BASELINE configs[1] through reference-shaped calls only, eager: get_samples_all -> pre-filter -> Renderer.render_batch_ray -> "
                       f"mapping_loss -> loss.backward() -> Adam([decoders, sdf table, colour table]).step(); {R} rays x {n_strat + n_imp} samples, "
                       f"{N_KEYFRAMES} keyframe pools, 2 hidden x {hidden} decoders ({prec})"}
    try:
        c2ws, pd, pc, pr = keyframe_pools(N_KEYFRAMES, bound, 5000, dev)
        n_per = R // N_KEYFRAMES
        cfg = {"rendering": {"perturb": True, "n_stratified": n_strat, "n_importance": n_imp}, "scale": 1, "grid_mode": "hash_grid",
               "grid": {"tcnn_network": False}, "model": {"mlp_precision": prec}}
        rend = us.Renderer(cfg, types.SimpleNamespace(bound=bound, device=dev, H=CAM["H"], W=CAM["W"], fx=CAM["fx"], fy=CAM["fy"], cx=CAM["cx"], cy=CAM["cy"]))

        def build(opt_kind):
            torch.manual_seed(0)
            dec = us.Decoders(cfg, c_dim=32, hidden_size=hidden, truncation=0.06, n_blocks=2).to(dev)
            es, ec = mk(16), mk(19)
            groups = [{"params": list(dec.parameters()), "lr": 0}, {"params": [es.params], "lr": 0}, {"params": [ec.params], "lr": 0}]
            opt = us.optim.Adam(groups) if opt_kind == "fused" else torch.optim.Adam(groups, **({"fused": True} if opt_kind == "torch_fused_kwarg" else {}))   # src/Mapper.py:118-121
            opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], opt.param_groups[2]["lr"] = LR["decoders"], LR["sdf_grid"], LR["color_grid"]   # :123-126
            return dec, es, ec, opt

        def iteration(dec, es, ec, opt, compact):
            H, Wd = CAM["H"], CAM["W"]
            opt.zero_grad()
            ro, rd, gd, gc = us.common.get_samples_all(0, H, 0, Wd, n_per, H, Wd, CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], c2ws, pd, pc, dev, pr)
            inside = us.common.bbox_filter(ro, rd, gd, bound)                             # src/Mapper.py:396-402
            valid = None
            if compact:
                ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]          # :403-406
            else:
                valid = inside
            ret = rend.render_batch_ray(([es], [ec]), dec, rd, ro, dev, 0.06, gt_depth=gd)
            loss = us.mapping_loss(ret, gd, gc, 0.06, W, valid=valid)
            loss.backward()
            opt.step()
            return loss.detach()

        for tag, opt_kind, compact in (("torch_adam_compaction", "torch", True), ("torch_adam_valid_flag", "torch", False),
                                       ("torch_adam_fused_kwarg_compaction", "torch_fused_kwarg", True),
                                       ("fused_adam_compaction", "fused", True), ("fused_adam_valid_flag", "fused", False)):
            m = build(opt_kind)
            ms, loss, ev = step_stats(lambda: iteration(*m, compact), steps, warmup)
            out[tag] = {"ms_per_iter": ms, "rays_per_s": R / (ms / 1e3), "final_loss": float(loss), "step_time_hip_events": ev}
            del m
        # the module seam ALONE (`import unislam_amd.tcnn as tcnn`, three changed lines): everything around the two tcnn
        # modules is the reference's own torch code (_TcnnOnlyModel above) -- ~60 torch launches per iteration around four module calls
        import unislam_amd.tcnn as tcnn
        res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)
        for tag, opt_kind in (("tcnn_only", "torch"), ("tcnn_only_fused_adam", "fused")):
            torch.manual_seed(0)
            m = _tcnn_only_model_class()(tcnn, hidden, prec, (16, 19), res, bound.to(dev), 0.06, W, n_strat, n_imp).to(dev)
            groups = [{"params": list(m.sdf_decoder.parameters()) + list(m.color_decoder.parameters()) + [m.beta], "lr": LR["decoders"]},
                      {"params": [m.enc_s.params], "lr": LR["sdf_grid"]}, {"params": [m.enc_c.params], "lr": LR["color_grid"]}]
            opt = us.optim.Adam(groups) if opt_kind == "fused" else torch.optim.Adam(groups)
            ms, loss, ev = step_stats(lambda: m.iteration(opt, c2ws, pd, pc, pr, n_per), steps, warmup)
            out[tag] = {"ms_per_iter": ms, "rays_per_s": R / (ms / 1e3), "final_loss": float(loss), "step_time_hip_events": ev}
            del m, opt
        out["ms_per_iter"] = out["torch_adam_compaction"]["ms_per_iter"]
        out["rays_per_s"] = out["torch_adam_compaction"]["rays_per_s"]
        out["note"] = ("headline of this block = torch_adam_compaction: the reference's lines unchanged except the imports; fused_adam_* swaps "
                       "torch.optim.Adam for unislam_amd.optim.Adam (same constructor, one launch per step); *_valid_flag hands the pre-filter's mask "
                       "to the loss instead of compacting four tensors (no host synchronisation); tcnn_only = ONLY `import unislam_amd.tcnn as tcnn` "
                       "(INTEGRATION.md 1): the reference's own torch renderer / sampler / losses around tcnn.Encoding + tcnn.Network, each encoder "
                       "and decoder its own autograd node (counted forward + scan + binned table gradient on a cached workspace, row-major features)")
    except Exception as e:                                # report, do not hide
        out["error"] = repr(e)[:400]
    return out


SCANNET_BOUND = [[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]]          # configs/ScanNet/scene0000.yaml:3
SCANNET_CAM = dict(H=460, W=620, fx=577.590698, fy=578.729797, cx=308.702667, cy=232.809998)   # scannet.yaml:37-44 after crop_edge 10


def config3_bench(us, dev, prec, steps, warmup):
    """
    BASELINE configs[2]: ScanNet scene0000, 8192 rays x 96 samples (80 stratified + 16 surface), tables log2T 16 / 16 at 2 cm (res 456),
    lr 0.02 (configs/ScanNet/scannet.yaml:12-18,25,46-47), the reference's ScanNet decoders (2 hidden x 16 with bias,
    src/networks/decoders.py:74-84), uncertainty-gated loss (m_mask_mode original), 25 % of the pool pixels WITHOUT a depth: those rays take
    the importance-sampling branch of src/utils/Renderer.py:104-130 every iteration -- its row count stays on the device, so the
    iteration is replayed from one hipGraph like the headline.  16 keyframes x 512 pixels; poses fixed, and with joint_opt.
    """
    out = {"workload": "BASELINE configs[2]: ScanNet scene0000 geometry, 8192 rays x 96 samples (80 + 16), L=16 F=2 tables log2T 16 / 16 res 456, 2 hidden x 16 "
                       "decoders with bias, uncertainty-gated loss, 25 % of the pixels without a depth (zero-depth branch in every iteration, "
                       "row count on the device), dense Adam; 16 keyframe pools x 512 pixels drawn inside the replayed graph"}
    try:
        global CAM
        bound = load_bound(SCANNET_BOUND)
        res = int((bound[:, 1] - bound[:, 0]).max() / 0.02)
        pls = per_level_scale(res)
        R, ns, ni, b = 8192, 80, 16, N_KEYFRAMES
        S, N = ns + ni, 8192 * 96
        cam_was, CAM = CAM, SCANNET_CAM
        try:
            c2ws, pd, pc, pr = keyframe_pools(b, bound, 4000, dev)
        finally:
            CAM = cam_was
        g = torch.Generator().manual_seed(7)
        holes = (torch.rand(pd.shape, generator=g) < 0.25).to(dev)
        pd = torch.where(holes, torch.zeros_like(pd), pd)

        def build():
            torch.manual_seed(0)
            mk = lambda: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 16,
                                                 "base_resolution": 16, "per_level_scale": pls}).to(dev)
            cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": prec}}
            dec = us.Decoders(cfg, c_dim=32, hidden_size=16, truncation=0.06, n_blocks=2).to(dev)
            return us.MapStep(mk(), mk(), dec, bound, ns, ni, 0.06, W, dict(decoders=0.001, sdf_grid=0.02, color_grid=0.02), mask_mode="original",
                              max_rays=R)

        def timed(fn):
            ms_, loss_, ev_ = step_stats(fn, steps, warmup)
            return ms_, float(loss_), ev_

        step = build()
        win = us.MapWindow(step, c2ws, pd, pc, pr, R // b, joint_opt=False, has_zero_depth=None)
        win.capture()
        ms, loss, ev = timed(win.replay)
        out["step_time_hip_events"] = ev
        n0 = int(step.zd_count)
        out.update({"mapping_iter_ms": ms, "rays_per_s": R / (ms / 1e3), "samples_per_s": N / (ms / 1e3), "final_loss": loss, "rays": R,
                    "samples_per_ray": S, "zero_depth_rays_last_iteration": n0, "n_params": int(step.n_flat),
                    "launch": "hipGraph replay of MapWindow (pixel draw + sampling + zero-depth branch + iteration in one graph)"})
        step.probe, step.probe_every, step._it = {}, 1, 0
        for _ in range(10):
            win.iterate()
        torch.cuda.synchronize()
        kern = {k: _median([a.elapsed_time(c) for a, c in v]) for k, v in step.probe.items()}
        step.probe = None
        if "hashgrid_bwd_joint" in kern and "hashgrid_scan_joint" in kern:
            kern["hashgrid_bwd_joint"] += kern.pop("hashgrid_scan_joint")
        out["kernel_ms"] = {k: round(v, 4) for k, v in sorted(kern.items())}
        alg = {"hashgrid_fwd_joint": 2 * 1024 * N, "hashgrid_bwd_joint": 2 * 2048 * N}
        out["roofline"] = {k: {"bound": "hbm", "achieved": alg[k] / (kern[k] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": alg[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg[k],
                               "avg_launch_ms": kern[k], "traffic": None} for k in alg if k in kern}
        out["roofline_note"] = ("algorithmic bytes / HBM peak, SERVED FROM L2 / Infinity Cache: both ScanNet tables are 6.5 MB (13 MB together, cache-resident), so "
                                "a frac beyond the achievable HBM rate (6.3 of 8 TB/s) is no HBM claim -- the gathers and the gradient sweep never reach HBM")
        out["algorithmic_table_bytes_per_step"] = sum(alg.values())
        step2 = build()
        win2 = us.MapWindow(step2, c2ws, pd, pc, pr, R // b, joint_opt=True, cam_lr=1e-3, has_zero_depth=None)
        win2.capture()
        ms2 = timed(win2.replay)[0]
        out["joint_opt_iteration_ms"] = ms2
        pd_full = torch.where(holes, torch.ones_like(pd), pd)
        step3 = build()
        win3 = us.MapWindow(step3, c2ws, pd_full, pc, pr, R // b, joint_opt=False, has_zero_depth=False)
        win3.capture()
        out["without_depth_holes_ms"] = timed(win3.replay)[0]
    except Exception as e:                                # report, do not hide
        out["error"] = repr(e)[:400]
    return out


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    global torch
    wd = float(os.environ.get("US_BENCH_WATCHDOG", "0"))            # seconds; > 0: a rank that is still running then dumps the Python
    if wd > 0:                                                     # stacks of all its threads to stderr (where does a stuck rank wait?)
        import faulthandler
        faulthandler.dump_traceback_later(wd, repeat=False, file=sys.stderr)
    # stdout carries ONE JSON line and nothing else: file descriptor 1 is pointed at stderr for the whole run (RCCL prints a version banner to
    # stdout when its first communicator comes up, other libraries may chat too), the record is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch as _torch
    torch = _torch
    import unislam_amd as us
    from unislam_amd.dist import init_from_env, broadcast_parameters
    # US_BENCH_REHEARSE=1: rehearse the N > 1 code path on a one-GPU box -- every rank on cuda:0, gloo as the transport (RCCL refuses
    # two ranks on one device).  The numbers of such a run mean nothing; it shows that the ranks start, step, agree and report.
    rehearse = os.environ.get("US_BENCH_REHEARSE") == "1"
    rank, local, world = init_from_env("gloo" if rehearse else None, timeout_s=float(os.environ.get("US_BENCH_PG_TIMEOUT", "180")))
    if rehearse:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X (unislam_amd has no CPU path)")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    n_strat, n_imp = 48, 16
    bound = load_bound(ROOM0_BOUND)
    res = int((bound[:, 1] - bound[:, 0]).max() / 0.01)                                  # 816 (src/UNISLAM.py:192-199)
    pls = per_level_scale(res)
    torch.manual_seed(0)
    mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
                                            "log2_hashmap_size": l2, "base_resolution": 16, "per_level_scale": pls}).to(dev)
    # N > 1 default: plain fp32 all-reduce in two announced segments, the colour table's hidden behind the sdf branch
    sharded, comm = args.sharded_adam, args.grad_comm
    if sharded:
        comm = "fp32"                                                                    # (the reduce-scatter works in place on the fp32 buffer)

    def build_step(prec, table_std=None, group="auto", dp_mode=None, sharded_adam=None, tcnn=False, hidden=None):
        torch.manual_seed(0)
        cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": bool(tcnn)}, "model": {"mlp_precision": prec}}
        dec = us.Decoders(cfg, c_dim=32, hidden_size=args.hidden if hidden is None else hidden, truncation=0.06, n_blocks=2).to(dev)
        es, ec = mk(16), mk(19)                                                          # replica.yaml:29-30
        if table_std is not None:                                                        # "trained-like" N(0, 0.1) tables
            with torch.no_grad():
                es.params.normal_(0.0, table_std); ec.params.normal_(0.0, table_std)
        st = us.MapStep(es, ec, dec, bound, n_strat, n_imp, 0.06, W, LR, max_rays=args.rays,
                        group=(True if world > 1 else None) if group == "auto" else group, bwd_mode=args.bwd_mode,
                        overlap=False if args.no_overlap else None, grad_comm="fp32" if sharded_adam else comm,
                        sharded_adam=sharded if sharded_adam is None else sharded_adam,
                        joint=None if args.joint == "auto" else args.joint == "1", dp_mode=dp_mode or args.dp_mode)
        st.decoder_pair = not args.no_decoder_pair
        return st, es, ec, dec

    if args.side == "dp_rank_local":                        # (a child of the N = 1 run: side_process)
        out = dp_rank_local_bench(us, lambda **kw: build_step(args.mlp_precision, **kw), bound, dev, args.steps, args.warmup)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        return

    step, es, ec, dec = build_step(args.mlp_precision)
    if world > 1:
        broadcast_parameters(step.flat)

    # ---- inputs: a fresh batch per step from the keyframe pools (src/Mapper.py:379-393), gathered into static tensors
    R = args.rays
    fresh = not args.fixed_batch and R % N_KEYFRAMES == 0
    use_graph = not args.no_graph and (world == 1 or fresh)      # N > 1: the rank-local launches as hipGraph segments between the collectives
    f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    if fresh:
        c2ws, pool_d, pool_c, pool_dirs = keyframe_pools(N_KEYFRAMES, bound, 1000 + rank, dev)
        P, n_per = pool_d.shape[1], R // N_KEYFRAMES

    launch_kind = {"graph": use_graph}

    def make_runner(st):
        """returns (next_step() -> loss, static ray tensors): one call = pixel draw + ray assembly + iteration.  Fresh batches: a
        window.MapWindow over the keyframe pools with the poses held fixed (joint_opt off) -- the draw, the gather + rotation and the
        sampling are ONE launch inside the replayed graph (us_window_sample)."""
        if fresh:
            win = us.MapWindow(st, c2ws, pool_d, pool_c, pool_dirs, n_per, joint_opt=False, has_zero_depth=False)
            graph = use_graph
            if graph:
                try:
                    win.capture(**({"collectives": args.dp_graph} if (world > 1 and not rehearse) else {}))   # N > 1: one graph, RCCL calls inside
                except Exception as e:                           # a runtime that refuses the capture: the eager step is the same arithmetic
                    print(f"[bench rank {rank}] graph capture failed, running eagerly: {e!r}"[:400], file=sys.stderr, flush=True)
                    graph = False
                if world > 1:
                    from unislam_amd.dist import all_agree as _agree
                    graph = _agree(graph, True, dev)
            launch_kind["graph"] = graph
            return (win.replay if graph else win.iterate), (win.ro, win.rd, win.gd, win.gc)
        ins = st.capture(R) if use_graph else (f32(R, 3), f32(R, 3), f32(R), f32(R, 3))
        for dst, src in zip(ins, synthetic_rays(R, bound, 1000 + rank, dev)):
            dst.copy_(src)
        return (st.replay if use_graph else (lambda: st.iterate(ins[0], ins[1], ins[2], ins[3], has_zero_depth=False))), ins

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    nxt, ins = make_runner(step)
    # set-up, before the W warm-up steps: keep the GPU busy for a fixed wall time so that the timed region does not start on a card
    # that is still ramping its clocks after the CPU-side scene construction (a 20-step region lasts 14 ms)
    # (every iteration of a data-parallel run contains collectives: the ranks must agree on how many they run, so the decision to go on
    #  is itself reduced over the ranks -- a wall-clock test per rank let them part ways and wait for each other forever)
    from unislam_amd.dist import all_agree as _all_agree
    all_agree = lambda flag: _all_agree(flag, True if world > 1 else None, dev)
    t_pre = time.perf_counter()
    while all_agree(time.perf_counter() - t_pre < args.prewarm_s):
        for _ in range(20):
            nxt()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        loss = nxt()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):                            # the timed region: K plain iterations, nothing else
        loss = nxt()
    barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = float(t.item())
    ms = 1e3 * el / args.steps

    # per-step GPU time of the same runner, one HIP event pair per step on the launch stream (median, p10, p90), AFTER the timed
    # region (the K timed steps above are bracketed by nothing but the barrier + synchronize of the contract)
    ev_stats = None
    if world == 1:
        n_ev = max(50, args.steps)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for a_, b_ in evs:
            a_.record(); nxt(); b_.record()
        torch.cuda.synchronize()
        ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        q = lambda f: ts[min(len(ts) - 1, int(f * len(ts)))]
        ev_stats = {"n": n_ev, "median_ms": q(0.5), "p10_ms": q(0.1), "p90_ms": q(0.9), "min_ms": ts[0], "max_ms": ts[-1],
                    "note": "HIP events around every step (graph launch to graph end on the launch stream), a separate pass after the timed region"}

    kern = None
    if not args.no_probe:
        # ---- the kernels' own durations: a SEPARATE pass after the timed region (every rank runs it: the iterations hold the
        #      collectives), eager, both branches on one stream, HIP events (torch.cuda.Event on the stream the kernels are
        #      launched on) around every hot launch
        step.probe, step.probe_every, step._it = {}, 1, 0
        eager = us.MapWindow(step, c2ws, pool_d, pool_c, pool_dirs, n_per, joint_opt=False, has_zero_depth=False).iterate if fresh else \
            (lambda: step.iterate(ins[0], ins[1], ins[2], ins[3], has_zero_depth=False))
        for _ in range(max(1, args.probe_steps)):
            eager()
        torch.cuda.synchronize()
        kern = {k: _median([a.elapsed_time(b) for a, b in v]) for k, v in step.probe.items()}
        step.probe = None

    if rank == 0:
        S = n_strat + n_imp
        N = R * S
        comm_desc = ("reduce-scatter + sharded Adam + all-gather" if sharded else "all-reduce") + \
                    f" of {(4 * step.n_flat if comm == 'fp32' else 2 * step.n_flat if comm == 'bf16' else 4 * step.o_tab_c + 2 * (step.n_flat - step.o_tab_c)) / 1e6:.1f} MB {comm} grads per step" + (f", dp_mode {args.dp_mode}" if world > 1 else "")
        rec = {"metric": "rays/s (64 samples, L=16 hash, 2x32 MLP), Replica room0 mapping iteration",
               "value": world * R / (ms / 1e3), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None,
               "dtype": "f32" if args.mlp_precision == "fp32" else "bf16 MFMA operands in the decoders (split hi + lo in the forward products), f32 tables, "
                                                                   "accumulation, parameters and optimiser",
               "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: Replica room0, 4096 rays x 64 samples (48 stratified + 16 surface), "
                                      "L=16 F=2 hash grids log2T 16 (sdf) / 19 (colour) res 816, 2 hidden x %d MLP decoders with bias, "
                                      "mapping iteration = pixel draw+ray gather+sample+encode+decode+composite+loss+backward+dense Adam, camera poses "
                                      "fixed (joint_opt off: the iteration of the first five keyframes, src/Mapper.py:519).  The reference's STEADY STATE "
                                      "is the joint_opt iteration (configs/UNISLAM.yaml:50: its default), which also optimises the window's poses: "
                                      "timed in the same run as `steady_state_joint_opt` (~15 %% slower); Replica's own 1 x 16 tcnn-layout f16 decoders: "
                                      "`replica_tcnn_decoders`; the reference's own call sequence under autograd: `drop_in_api`" % args.hidden,
                          "rays_per_gpu": R, "samples_per_ray": S, "points_per_gpu": N, "n_params": int(step.n_flat),
                          "batch": (f"fresh per step: {n_per} pixels from each of {N_KEYFRAMES} keyframe pools of {P} pixels, drawn inside the step (us_window_sample: "
                                    f"counter-based draw + gather + rotation + sampling, one launch of the replayed graph)"
                                    if fresh else "one fixed batch re-rendered every step"),
                          "parallelism": f"dp{world} (frames/rays sharded, {comm_desc})"},
               "rays_per_s_per_gpu": R / (ms / 1e3), "mapping_iter_ms": ms, "final_loss": float(loss), "step_time_hip_events": ev_stats,
               "launch": (("hipGraph replay of MapWindow.iterate (joint_opt off): pixel draw + gather + sampling in one launch, then MapStep.iterate"
                           if world == 1 else "hipGraph segments of MapWindow.iterate between the collectives (statistics all-reduce, gradient segments)")
                          if (use_graph and fresh and launch_kind["graph"]) else "hipGraph replay of MapStep.iterate" if (use_graph and not fresh) else "eager")}
        if kern is not None:
            rec["kernel_ms"] = {k: round(v, 4) for k, v in sorted(kern.items())}
            rec["kernel_ms_note"] = f"median over {max(1, args.probe_steps)} eager one-stream iterations after the timed region"
            # algorithmic bytes per launch: forward gather 16 levels x 8 corners x 2 feat x 4 B = 1024 B/point/grid,
            # backward scatter counted read+write = 2048 B/point/grid; a launch that serves both grids moves both grids' bytes
            alg = {"hashgrid_fwd_sdf": 1024 * N, "hashgrid_fwd_color": 1024 * N, "hashgrid_bwd_sdf": 2048 * N, "hashgrid_bwd_color": 2048 * N,
                   "hashgrid_fwd_joint": 2 * 1024 * N, "hashgrid_bwd_joint": 2 * 2048 * N}
            if "hashgrid_bwd_joint" in kern and "hashgrid_bwd_joint_sdf" in kern:      # N > 1: the accumulate pass is split per grid
                kern["hashgrid_bwd_joint"] += kern.pop("hashgrid_bwd_joint_sdf")
            crit = None
            if "hashgrid_bwd_joint" in kern and "hashgrid_scan_joint" in kern:
                # the scan passes and the item table of the joint table gradient run beside the decoders; they belong to the gradient.
                # What the replayed iteration waits for is the gradient call itself (record pass + accumulate): reported beside it.
                crit = kern["hashgrid_bwd_joint"]
                kern["hashgrid_bwd_joint"] += kern.pop("hashgrid_scan_joint")
                rec["kernel_ms"] = {k: round(v, 4) for k, v in sorted(kern.items())}
                rec["kernel_ms_note"] += "; hashgrid_bwd_joint = scans + item table (issued in the forward phase) + record pass + accumulate"
            alg = {k: v for k, v in alg.items() if k in kern}
            dom = max(alg, key=lambda k: kern[k])
            ach = alg[dom] / (kern[dom] * 1e-3) / 1e9
            traffic, tsrc = None, None
            tj = os.environ.get("US_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "traffic.json"))
            if os.path.exists(tj):
                tr = json.load(open(tj))
                traffic, tsrc = tr.get(dom), "profile-derived (not measured in this run): " + os.path.relpath(tj, ROOT) + " -- " + tr.get("_note", "")
            rec["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                               "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kern[dom]}
            if crit is not None and dom == "hashgrid_bwd_joint":
                rec["roofline"]["critical_path"] = {"what": "record pass + accumulate (the scans and the item table run beside the decoders on their own stream)",
                                                    "avg_launch_ms": crit, "achieved": alg[dom] / (crit * 1e-3) / 1e9,
                                                    "frac": alg[dom] / (crit * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the other table kernels beside the dominant one (the encoder is the largest SINGLE kernel of the iteration)
            rec["roofline_by_kernel"] = {k: {"achieved": alg[k] / (kern[k] * 1e-3) / 1e9, "frac": alg[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "unit": "GB/s", "algorithmic_bytes_per_launch": alg[k], "avg_launch_ms": kern[k]} for k in alg}
            if "hashgrid_fwd_joint" in rec["roofline_by_kernel"]:
                try:
                    rec["roofline_by_kernel"]["hashgrid_fwd_joint"]["l2_request"] = encoder_request_roofline((step.es.desc, step.ec.desc), step.pts[:R],
                                                                                                             kern["hashgrid_fwd_joint"])
                except Exception as e:                         # report, do not hide
                    rec["roofline_by_kernel"]["hashgrid_fwd_joint"]["l2_request"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_extras:
            # also the forward-only rate (the render_img / meshing use) and the iteration without Adam
            def timed(fn, k=max(10, args.steps // 2)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(k):
                    fn()
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t1) / k
            render = lambda: step.forward(ins[0], ins[1], ins[2], ins[3], has_zero_depth=False, backward_follows=False)
            fwd_ms = timed(render)
            fb_ms = timed(lambda: step.forward_backward(ins[0], ins[1], ins[2], ins[3], has_zero_depth=False))
            rec["extra"] = {"forward_only_ms": fwd_ms, "forward_only_rays_per_s": R / (fwd_ms / 1e3),
                            "iteration_without_adam_ms": fb_ms}

            def side_run(st):
                n2, _ = make_runner(st)
                m2, l2, ev2 = step_stats(n2, args.steps, args.warmup)
                return {"ms_per_step": m2, "rays_per_s": R / (m2 / 1e3), "final_loss": float(l2), "step_time_hip_events": ev2}
            if args.mlp_precision != "fp32":
                # the same iteration with f32-input MFMA decoders (exact fmaf chains, what the fixture comparisons run)
                rec["fp32_decoders"] = side_run(build_step("fp32")[0])
            # tables with "trained-like" N(0, 0.1) entries: alpha is no longer degenerate, the gradients are dense in value
            rec["trained_like_tables"] = side_run(build_step(args.mlp_precision, table_std=0.1)[0])
            # Replica's OWN decoder configuration (configs/Replica/replica.yaml:33 grid.tcnn_network True -> src/networks/decoders.py:50-70):
            # FullyFusedMLP 32 -> 16 -> out, ONE hidden layer, no bias, half-precision arithmetic (f16 MFMA here), tcnn's flat parameter
            # layout -- SURVEY 8d's "ref-exact (16-wide)" run of cfg2 beside the "2x32" headline
            try:
                r_ = side_run(build_step("f16", tcnn=True, hidden=16)[0])
                r_["decoders"] = "tcnn layout (sdf_decoder.params / color_decoder.params), 1 hidden x 16, no bias, f16 MFMA operands (us_mlp_desc US_PREC_F16)"
                rec["replica_tcnn_decoders"] = r_
            except Exception as e:                            # report, do not hide
                rec["replica_tcnn_decoders"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_extras:
            rec["joint_opt"] = joint_opt_bench(us, lambda: build_step(args.mlp_precision), bound, dev, args.steps, args.warmup)
            jo = rec["joint_opt"].get("4096_rays_16_keyframes", {})
            if "joint_opt_iteration_ms" in jo:
                # the reference's STEADY-STATE iteration (joint_opt: True is its default, configs/UNISLAM.yaml:50, active from the fifth keyframe,
                # src/Mapper.py:519): the same batch with the window's poses optimised along -- promoted to the top level beside `value`
                rec["steady_state_joint_opt"] = {"ms_per_step": jo["joint_opt_iteration_ms"], "rays_per_s": jo["rays_per_s"],
                                                 "what": "the headline's batch (4096 rays x 64 over 16 keyframes) with joint_opt on: pose -> rays, the pose "
                                                         "gradients and the poses' Adam group inside the iteration; `value` above is the iteration "
                                                         "with the poses fixed (the first five keyframes of a run)"}
        if world == 1 and not args.no_extras:
            rec["drop_in_api"] = drop_in_bench(us, dev, args.mlp_precision, args.hidden, bound, mk, args.steps, args.warmup, R, n_strat, n_imp)
        if world == 1 and not args.no_extras:
            rec["config3"] = config3_bench(us, dev, args.mlp_precision, args.steps, args.warmup)
        if world == 1 and not args.no_tracking:
            rec["tracking"] = tracking_bench(us, es, ec, dec, bound, dev)
        if world == 1 and not args.no_tracking and not args.no_extras:
            rec["slam_frame"] = slam_bench(us, dev, args.hidden, args.mlp_precision)
        if world == 1 and not args.no_extras:
            rec["dp_rank_local_ms"] = side_process("dp_rank_local", args)     # (RCCL group + captured collectives: in a process of its own)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(bound, n_strat, n_imp, args.hidden, R)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
