#!/usr/bin/env python3
"""
bench.py -- throughput of Uni-SLAM's mapping iteration (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one mapping iteration of the hot path on one batch of synthetic rays already resident in HBM:
bbox pre-filter -> z sampling (+jitter) -> points -> 2x hash-grid encode -> 2x fused MLP -> SDF->alpha compositing ->
uncertainty-gated loss -> full backward (table, decoder and beta gradients) -> [all-reduce over ranks] -> Adam over all
12.9 M parameters.  Workload (config.workload): BASELINE configs[1] = Replica room0 geometry, 4096 rays x 64 samples
(48 stratified + 16 surface), L=16 F=2 hash grids (log2T 16 sdf / 19 colour, finest resolution 816), 2 hidden x 32
MLP decoders.  Weak scaling: every rank renders its own 4096 rays (one synthetic frame per rank, configs[3]).

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant kernel, HIP-event timed inside the
timed region) and `cpu_baseline` (the CPU oracle port timed on the host cores, rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROOM0_BOUND = [[-1.0, 7.0], [-1.3, 3.7], [-1.7, 1.4]]           # configs/Replica/room0.yaml:3
W = dict(fs=5, center=200, tail=10, color=5, depth=0.1)          # configs/UNISLAM.yaml:67-71
LR = dict(decoders=0.001, sdf_grid=0.05, color_grid=0.05)        # configs/Replica/replica.yaml:19-21
HBM_PEAK_GBS = 8000.0                                            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def load_bound(bound, dividable=0.24):
    """src/UNISLAM.py:205-218"""
    b = torch.tensor(bound, dtype=torch.float64).float()
    b[:, 1] = (((b[:, 1] - b[:, 0]) / dividable).int() + 1) * dividable + b[:, 0]
    return b


def per_level_scale(res, n_levels=16):
    return float(2.0 ** (math.log2(res / n_levels) / (n_levels - 1)))      # src/UNISLAM.py:241


def synthetic_rays(R, bound, seed, device):
    """a camera at the scene centre with a seeded random rotation, Replica intrinsics, U(0.5,3.5) m depths"""
    g = torch.Generator().manual_seed(seed)
    H, Wd, fx, fy, cx, cy = 680, 1200, 600.0, 600.0, 599.5, 339.5
    q = torch.randn(4, generator=g); q = q / q.norm()
    r, i, j, k = q.tolist()
    Rm = torch.tensor([[1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r)],
                       [2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r)],
                       [2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)]])
    u = torch.randint(Wd, (R,), generator=g).float(); v = torch.randint(H, (R,), generator=g).float()
    dirs = torch.stack([(u - cx) / fx, -(v - cy) / fy, -torch.ones(R)], -1)
    rays_d = dirs @ Rm.t()
    centre = bound.mean(dim=1)
    rays_o = centre.expand(R, 3).contiguous()
    t = (bound.unsqueeze(0) - rays_o.unsqueeze(-1)) / rays_d.unsqueeze(-1)
    far = torch.min(torch.max(t, dim=2)[0], dim=1)[0]
    depth = torch.minimum(torch.rand(R, generator=g) * 3.0 + 0.5, 0.9 * far)
    color = torch.rand(R, 3, generator=g)
    return rays_o.to(device), rays_d.to(device), depth.to(device), color.to(device)


def cpu_baseline(bound, n_strat, n_imp, hidden, budget_s=15.0):
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
    R = 512
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
    H, Wd, fx, fy, cx, cy = 680, 1200, 600.0, 600.0, 599.5, 339.5
    g = torch.Generator().manual_seed(5)
    gt_depth = (torch.rand(1, H, Wd, generator=g) * 1.5 + 0.8).to(dev)
    gt_color = torch.rand(1, H, Wd, 3, generator=g).to(dev)
    centre = bound.mean(dim=1)
    pose = torch.tensor([[0.9, 0.1, -0.2, 0.3, float(centre[0]), float(centre[1]), float(centre[2])]], device=dev)
    quad = torch.nn.Parameter(pose[:, :4].clone()); T = torch.nn.Parameter(pose[:, 4:].clone())
    opt = torch.optim.Adam([{"params": [T], "lr": 2e-3, "betas": (0.5, 0.999)}, {"params": [quad], "lr": 1e-3, "betas": (0.5, 0.999)}],
                           capturable=True)
    ts = us.TrackStep(es, ec, dec, bound, 32, 8, 0.06, dict(fs=10, center=200, tail=50, color=5, depth=1), max_rays=2000)
    step = lambda: ts.iterate(torch.cat([quad, T], -1), gt_color, gt_depth, 2000, opt, H, Wd, fx, fy, cx, cy, 75, 75)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss, _, _ = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    out = {"workload": "Replica tracking iteration: 2000 rays x 40 samples, pose Adam", "eager_ms_per_iter": ms,
           "eager_rays_per_s": 2000 / (ms / 1e3), "iters": iters, "final_loss": float(loss)}
    try:                                                  # the same iteration captured into a hipGraph and replayed
        it = us.CapturedIteration(step)
        for _ in range(10):
            it.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss, _, _ = it.replay()
        torch.cuda.synchronize()
        gms = 1e3 * (time.perf_counter() - t0) / iters
        out.update({"graph_ms_per_iter": gms, "graph_rays_per_s": 2000 / (gms / 1e3), "graph_final_loss": float(loss)})
    except Exception as e:                                # report, do not hide
        out["graph_error"] = repr(e)[:300]
    try:                                                  # fully fused variant: pose->rays, pose gradient, pose Adam as HIP kernels
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
        out.update({"fused_eager_ms_per_iter": fms})
        it = us.CapturedIteration(fstep)
        for _ in range(10):
            it.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss, _, _ = it.replay()
        torch.cuda.synchronize()
        fg = 1e3 * (time.perf_counter() - t0) / iters
        out.update({"fused_graph_ms_per_iter": fg, "fused_graph_rays_per_s": 2000 / (fg / 1e3), "fused_final_loss": float(loss)})
    except Exception as e:
        out["fused_error"] = repr(e)[:300]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--hidden", type=int, default=32, help="MLP width (32 = BASELINE '2x32'; 16 = reference decoders.py default)")
    ap.add_argument("--bwd-mode", type=int, default=-1)
    ap.add_argument("--mlp-precision", default="fp32", choices=["fp32", "bf16"],
                    help="MFMA operand type of the decoders; the headline (parity-tested to 1e-3) is fp32")
    ap.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16"],
                    help="payload type of the gradient all-reduce (N > 1); bf16 halves the xGMI bytes, not bit-faithful to one process")
    ap.add_argument("--sharded-adam", action="store_true",
                    help="N > 1: reduce-scatter the gradient, Adam on this rank's shard, all-gather the parameters")
    ap.add_argument("--no-overlap", action="store_true", help="run the sdf and colour branches on one stream")
    ap.add_argument("--packed-records", action="store_true", help="8-byte intermediate records in the table gradient (US_GRID_BWD_PACKED)")
    ap.add_argument("--no-graph", action="store_true", help="N = 1: launch every iteration eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-tracking", action="store_true")
    args = ap.parse_args()

    import unislam_amd as us
    from unislam_amd.dist import init_from_env, broadcast_parameters
    # US_BENCH_REHEARSE=1: rehearse the N > 1 code path on a one-GPU box -- every rank on cuda:0, gloo as the transport (RCCL refuses
    # two ranks on one device).  The numbers of such a run mean nothing; it shows that the ranks start, step, agree and report.
    rehearse = os.environ.get("US_BENCH_REHEARSE") == "1"
    rank, local, world = init_from_env("gloo" if rehearse else None)
    if rehearse:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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

    def build_step(prec):
        torch.manual_seed(0)
        cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": prec}}
        dec = us.Decoders(cfg, c_dim=32, hidden_size=args.hidden, truncation=0.06, n_blocks=2).to(dev)
        es, ec = mk(16), mk(19)                                                          # replica.yaml:29-30
        st = us.MapStep(es, ec, dec, bound, n_strat, n_imp, 0.06, W, LR, max_rays=args.rays,
                        group=True if world > 1 else None, bwd_mode=args.bwd_mode, overlap=False if args.no_overlap else None,
                        grad_comm=args.grad_comm, sharded_adam=args.sharded_adam, packed_records=args.packed_records)
        return st, es, ec, dec

    step, es, ec, dec = build_step(args.mlp_precision)
    if world > 1:
        broadcast_parameters(step.flat)
    ro, rd, gd, gc = synthetic_rays(args.rays, bound, 1000 + rank, dev)                   # one synthetic frame per rank

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step.iterate(ro, rd, gd, gc, has_zero_depth=False)
    if not args.no_probe:
        step.probe = {}                 # every 10th timed step carries HIP events around the hot kernels (and runs its two
        step.probe_every = 10 if args.steps >= 50 else max(1, args.steps // 5)   # branches on one stream: the kernels' own durations)
        step._it = 0
    # single process: the iteration is replayed from a hipGraph (MapStep.capture; same kernels, one host call, the two branch streams
    # scheduled by the graph); every probe_every-th step runs eagerly with HIP events around its kernels, as before
    use_graph = world == 1 and not args.no_graph
    if use_graph:
        for dst, src in zip(step.capture(args.rays), (ro, rd, gd, gc)):
            dst.copy_(src)
        for _ in range(3):
            step.replay()
        pe = step.probe_every if step.probe is not None else 0
        step.probe_every = 1
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if use_graph and not (pe and k % pe == 0):
            loss = step.replay()
        else:
            loss = step.iterate(ro, rd, gd, gc, has_zero_depth=False)
    barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = float(t.item())
    ms = 1e3 * el / args.steps

    if rank == 0:
        S = n_strat + n_imp
        N = args.rays * S
        rec = {"metric": "rays/s (64 samples, L=16 hash, 2x32 MLP), Replica room0 mapping iteration",
               "value": world * args.rays / (ms / 1e3), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32" if args.mlp_precision == "fp32" else "f32 tables/accumulation, bf16 MFMA operands in the decoders",
               "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: Replica room0, 4096 rays x 64 samples (48 stratified + 16 surface), "
                                      "L=16 F=2 hash grids log2T 16 (sdf) / 19 (colour) res 816, 2 hidden x %d MLP decoders with bias, "
                                      "mapping iteration = sample+encode+decode+composite+loss+backward+dense Adam" % args.hidden,
                          "rays_per_gpu": args.rays, "samples_per_ray": S, "points_per_gpu": N, "n_params": int(step.n_flat),
                          "parallelism": f"dp{world} (frames/rays sharded, 1 all-reduce of {(4 if args.grad_comm == 'fp32' else 2) * step.n_flat / 1e6:.1f} MB "
                                         f"{args.grad_comm} grads per step)"},
               "rays_per_s_per_gpu": args.rays / (ms / 1e3), "mapping_iter_ms": ms, "final_loss": float(loss),
               "launch": "hipGraph replay of MapStep.iterate (every %d-th step eager with HIP-event probes)" % pe if use_graph else "eager"}
        if step.probe:
            kern = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in step.probe.items()}
            rec["kernel_ms"] = {k: round(v, 4) for k, v in sorted(kern.items())}
            # algorithmic bytes per launch: forward gather 16 levels x 8 corners x 2 feat x 4 B = 1024 B/point/grid,
            # backward scatter counted read+write = 2048 B/point/grid
            alg = {"hashgrid_fwd_sdf": 1024 * N, "hashgrid_fwd_color": 1024 * N, "hashgrid_bwd_sdf": 2048 * N, "hashgrid_bwd_color": 2048 * N}
            dom = max(alg, key=lambda k: kern.get(k, 0.0))
            ach = alg[dom] / (kern[dom] * 1e-3) / 1e9
            traffic = None
            tj = os.environ.get("US_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "traffic.json"))
            if os.path.exists(tj):
                traffic = json.load(open(tj)).get(dom)
            rec["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg[dom],
                               "avg_launch_ms": kern[dom]}
        if world == 1 and not args.no_probe:
            # also the forward-only rate (the render_img / meshing use) and the iteration without Adam
            step.probe = None
            def timed(fn, k=max(10, args.steps // 2)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(k):
                    fn()
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t1) / k
            fwd_ms = timed(lambda: step.forward(ro, rd, gd, gc, has_zero_depth=False))
            fb_ms = timed(lambda: step.forward_backward(ro, rd, gd, gc, has_zero_depth=False))
            rec["extra"] = {"forward_only_ms": fwd_ms, "forward_only_rays_per_s": args.rays / (fwd_ms / 1e3),
                            "iteration_without_adam_ms": fb_ms}
        if world == 1 and args.mlp_precision == "fp32" and not args.no_probe:
            # the same iteration with bf16 MFMA operands in the two decoders (v_mfma_f32_16x16x32_bf16); not the headline
            st2 = build_step("bf16")[0]
            for _ in range(args.warmup):
                st2.iterate(ro, rd, gd, gc, has_zero_depth=False)
            if use_graph:
                for dst, src in zip(st2.capture(args.rays), (ro, rd, gd, gc)):
                    dst.copy_(src)
                st2.replay()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(args.steps):
                l2 = st2.replay() if use_graph else st2.iterate(ro, rd, gd, gc, has_zero_depth=False)
            torch.cuda.synchronize()
            ms2 = 1e3 * (time.perf_counter() - t1) / args.steps
            rec["bf16_decoders"] = {"ms_per_step": ms2, "rays_per_s": args.rays / (ms2 / 1e3), "final_loss": float(l2)}
            del st2
        if world == 1 and not args.no_tracking:
            rec["tracking"] = tracking_bench(us, es, ec, dec, bound, dev)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(bound, n_strat, n_imp, args.hidden)
        print(json.dumps(rec), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
