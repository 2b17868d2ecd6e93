"""
GPU: the thin tracking / mapping drivers (unislam_amd.slam, after src/Tracker.py:271-370 and src/Mapper.py:177-545) on a synthetic
RGB-D sequence (unislam_amd.synthetic): the trajectory is recovered, tracking beats dead reckoning, the keyframe machinery and
the joint pose optimisation run.
"""
import numpy as np
import pytest
import torch

import unislam_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(us, n_frames, seed=0, mlp_precision="fp32", graph_replay=None, room=None, every=2):
    from unislam_amd.synthetic import SyntheticRoom
    from unislam_amd.slam import SLAM
    torch.manual_seed(seed)
    frames = SyntheticRoom(n_frames=n_frames, H=120, W=160, device=DEV, **(room or {}))
    bound = O.load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]])
    res = int((bound[:, 1] - bound[:, 0]).max() / 0.02)
    ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                       "base_resolution": 16, "per_level_scale": O.per_level_scale(res)}
    es, ec = us.HashGridEncoding(3, ecfg(16)).to(DEV), us.HashGridEncoding(3, ecfg(16)).to(DEV)
    cfg = {"rendering": {"perturb": True, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid",
           "grid": {"tcnn_network": False}, "model": {"mlp_precision": mlp_precision}}
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06).to(DEV)
    dec.bound = bound
    slam = SLAM(frames, (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy), es, ec, dec, bound,
                cfg={"tracking": dict(ignore_edge_W=8, ignore_edge_H=8, pixels=1000, iters=10),
                     "mapping": dict(dict(pixels=2000, iters=20, iters_first=300, every_frame=every, keyframe_every=every),
                                     **({} if graph_replay is None else dict(graph_replay=graph_replay)))})
    return slam, frames


def _g15_slam(us, g, seed=0, prec="fp32", P=None, draws=None):
    """the HIP drivers with fixture g15's (or, P = G16, g16's) scene, settings and initial decoders (oracle/g15_settings.py); draws: a draw
    source (slam.TorchDraws) -> also the fixture's initial TABLES (the generator of oracle/gen_golden.py _ref_loop, restated here)"""
    from g15_settings import G15
    P = G15 if P is None else P
    from unislam_amd.synthetic import SyntheticRoom
    from unislam_amd.slam import SLAM
    T_, M_ = P["tracking"], P["mapping"]
    torch.manual_seed(seed)
    frames = SyntheticRoom(n_frames=P["n_frames"], H=P["H"], W=P["W"], fov_deg=P["fov_deg"], device=DEV, tex_freq=P["tex_freq"])
    bound = O.load_bound(P["room_bound"])
    res = int((bound[:, 1] - bound[:, 0]).max() / P["voxel"])
    assert res == int(g["res"])
    ecfg = lambda l2: {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2, "base_resolution": 16,
                       "per_level_scale": O.per_level_scale(res)}
    es, ec = us.HashGridEncoding(3, ecfg(P["log2T"][0]), seed=seed + 1).to(DEV), us.HashGridEncoding(3, ecfg(P["log2T"][1]), seed=seed + 2).to(DEV)
    cfg = {"rendering": {"perturb": True, "n_stratified": P["n_stratified"], "n_importance": P["n_importance"]}, "scale": 1, "grid_mode": "hash_grid",
           "grid": {"tcnn_network": False}, "model": {"mlp_precision": prec}}
    dec = us.Decoders(cfg, c_dim=32, truncation=P["truncation"]).to(DEV)
    dec.load_state_dict({k[len("dec0__"):].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("dec0__")})
    dec.bound = bound
    w = lambda d: dict(fs=d["w_sdf_fs"], center=d["w_sdf_center"], tail=d["w_sdf_tail"], depth=d["w_depth"], color=d["w_color"])
    slam = SLAM(frames, (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy), es, ec, dec, bound,
                cfg={"tracking": dict({k: T_[k] for k in ("pixels", "iters", "lr_T", "lr_R", "ignore_edge_H", "ignore_edge_W", "const_speed_assumption",
                                                          "activated_mapping_mode", "uncertainty_ts")}, w=w(T_)),
                     "mapping": dict({k: M_[k] for k in ("pixels", "iters", "iters_first", "every_frame", "keyframe_every", "lr_first_factor", "lr_factor",
                                                         "joint_opt", "joint_opt_cam_lr", "mapping_window_size", "LC")},
                                     lr=dict(decoders=M_["decoders_lr"], sdf_grid=M_["hash_grids_lr"], color_grid=M_["c_hash_grids_lr"]), w=w(M_)),
                     "rendering": dict(n_stratified=P["n_stratified"], n_importance=P["n_importance"], perturb=True), "truncation": P["truncation"]},
                draws=draws)
    if draws is not None:
        gi = torch.Generator().manual_seed(P["seed"] + 1)
        with torch.no_grad():
            es.params.copy_(((torch.rand(es.params.shape, generator=gi) * 2 - 1) * 1e-4).to(DEV))
            ec.params.copy_(((torch.rand(ec.params.shape, generator=gi) * 2 - 1) * 1e-4).to(DEV))
    return slam, frames


class _CheckedDraws:
    """slam.TorchDraws on the generator state the fixture stores, every draw checked against the fixture's log of the REFERENCE loop's draws
    (oracle/gen_golden.py _DrawLog: kind, sizes, f64 sum): the replay consumes the reference's stream draw for draw, or says where it left it"""
    KINDS = ("randint", "rand", "randperm")

    def __init__(self, g):
        from unislam_amd.slam import TorchDraws
        self.src = TorchDraws(state=g["rng_state"])
        self.g, self.k, self.frame = g, 0, -1

    def _check(self, kind, a, b, out):
        g, k = self.g, self.k
        assert k < len(g["draw_kind"]), f"draw {k}: the reference's loop took {len(g['draw_kind'])} draws, the replay asks for more ({kind} {a} {b})"
        ref = (self.KINDS[int(g["draw_kind"][k])], int(g["draw_a"][k]), int(g["draw_b"][k]))
        assert ref == (kind, a, b), f"draw {k} (reference frame {int(g['draw_frame'][k])}): the reference drew {ref}, the replay asks for {(kind, a, b)}"
        assert abs(float(out.double().sum()) - float(g["draw_sum"][k])) <= 1e-9 * max(1.0, abs(float(g["draw_sum"][k]))), f"draw {k}: different numbers"
        self.k += 1
        return out

    def randint(self, high, n):
        return self._check("randint", int(high), int(n), self.src.randint(high, n))

    def rand(self, rows, cols):
        return self._check("rand", int(rows), int(cols), self.src.rand(rows, cols))

    def randperm(self, n):
        return self._check("randperm", int(n), 0, self.src.randperm(n))


def _replay_against(us, g, P):
    """run the HIP drivers on the fixture's stream; returns (slam, per-frame translation error against the REFERENCE's estimate, relative to
    the distance travelled so far + 10 cm)"""
    draws = _CheckedDraws(g)
    slam, frames = _g15_slam(us, g, P=P, draws=draws)
    assert abs(float(slam.es.params.detach().double().sum()) - float(g["table_sdf_sum"])) < 1e-9 and \
        abs(float(slam.ec.params.detach().double().sum()) - float(g["table_color_sum"])) < 1e-9           # the reference's initial tables
    slam.run()
    assert draws.k == len(g["draw_kind"]), (draws.k, len(g["draw_kind"]))                         # ... and took every draw of it
    est, ref = slam.estimate_c2w_list[:, :3, 3].cpu().double(), torch.from_numpy(g["est_c2w"][:, :3, 3]).double()
    # every iteration's loss, tracking and mapping, in the loop's order (the reference's: every scalar its loop called .backward() on)
    mine, theirs = np.array(slam.history["losses"]), g["loss_log"].astype(np.float64)
    assert mine.shape == theirs.shape, (mine.shape, theirs.shape)
    rel = np.abs(mine - theirs) / np.maximum(np.abs(theirs), 1e-6)
    first = np.nonzero(rel > 1e-3)[0]
    print(f"losses: {len(mine)} iterations; relative difference: first 60 {np.array2string(rel[:60], precision=1, max_line_width=250)}; "
          f"first beyond 1e-3: iteration {int(first[0]) if len(first) else None} (frame {int(g['loss_frame'][first[0]]) if len(first) else None})")
    return slam, (est - ref).norm(dim=-1)


def test_g15_sequence_against_the_reference_loop(golden):
    """
    BASELINE configs[4] ("full tracking + mapping loop ... ATE vs reference") at the size the CPU can drive the REFERENCE at: fixture
    g15_sequence is the reference's own loop (Tracker.run / Mapper.run bodies around its optimize_tracking / optimize_mapping /
    keyframe_selection_LC, oracle/gen_golden.py g15) over 34 frames of the analytic room, every frame tracked, mapped and kept as a keyframe:
    joint_opt from the fifth keyframe, the extra rays beyond 20 keyframes.  The HIP drivers run the same frames with the same settings and
    the same initial decoders.  The two draw different pixels (the kernels draw their own), so the trajectories are two samples of one
    process: the keyframe list must be the same; the ATE of every seed stays within 2.5 x the reference's, the median of three within
    1.25 x; the largest per-frame error within 3 x the reference's largest (measured over eight runs: ATE 2.2 - 5.5 cm against 3.24, largest
    error 3.0 - 8.7 cm against 4.4, always at frames 4 - 12 where the map is a few frames old).
    """
    import unislam_amd as us
    g = golden("g15_sequence")
    ref_ate, ref_max = float(g["ate_rmse_m"]), float(g["err_m"].max())
    assert 0.01 < ref_ate < 0.06 and len(g["keyframe_list"]) == 34 and int(g["joint_opt"].sum()) == 29       # the fixture is what its header says
    ates = []
    for seed in range(3):
        slam, frames = _g15_slam(us, g, seed=seed)
        if seed == 0:
            # the analytic frames the reference saw (rendered on the CPU there): same scene, same poses
            _, c0, d0, _, _ = frames[0]
            np.testing.assert_allclose(d0[frames.H // 2].cpu().numpy(), g["frame0_depth_row"], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(c0[frames.H // 2].cpu().numpy(), g["frame0_color_row"], rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(frames.poses.cpu().numpy(), g["gt_c2w"], atol=1e-6)
        slam.run()
        err = (slam.estimate_c2w_list[:, :3, 3] - slam.gt_c2w_list[:, :3, 3]).norm(dim=-1)
        ate = slam.ate_rmse()
        print(f"g15 seed {seed}: ATE {100 * ate:.2f} cm (reference loop {100 * ref_ate:.2f}), max {100 * float(err.max()):.2f} cm (reference {100 * ref_max:.2f})")
        assert slam.mapper.keyframe_list == [int(k) for k in g["keyframe_list"]]
        assert slam.mapper.joint_opt and slam.mapper.LC_cnt == int(g["lc_cnt"])
        kc = slam.mapper.kind_counts
        # first frame | poses fixed (keyframes 1..4) | joint_opt | joint_opt + extra rays (more than 20 keyframes): as often as in the reference's run
        assert kc[(False, False, False, 5.0)] == 1 and kc[(False, False, False, 1.0)] == 4
        assert kc[(True, False, False, 1.0)] == int(((g["keyframes_before_mapping"] > 4) & (g["keyframes_before_mapping"] <= 20)).sum())
        assert kc[(True, True, False, 1.0)] == int((g["keyframes_before_mapping"] > 20).sum()) == 13
        assert ate <= 2.5 * ref_ate and float(err.max()) <= 3.0 * ref_max, (ate, float(err.max()))
        ates.append(ate)
    assert sorted(ates)[1] <= 1.25 * ref_ate, ates


def test_g15_replayed_draw_for_draw(golden):
    """
    The loop against the reference's loop, DETERMINISTICALLY (r6): fixture g15 stores the state of torch's generator at the start of the
    reference's loop and a log of every draw it took (1493: pixel indices, jitter, keyframe pools).  The HIP drivers run with
    SLAM(draws=TorchDraws(state)) -- the same pixels, the same jitter, the same pools, checked draw for draw -- so the two runs differ by
    floating-point arithmetic only (CPU fp32 autograd + torch.optim.Adam there; HIP kernels here, fp32 decoders).  Measured: the first
    iteration's loss agrees to 1e-7, then the difference doubles per Adam step of the first mapped frame (lr 0.25 against tables
    initialised at 1e-4: Adam's first steps are +-lr * sign(g), discontinuous where a gradient nearly cancels) and saturates at 1e-2 by
    iteration 20: from there on the two runs are two samples of one process, 3 - 20 mm apart per frame (both 30 - 40 mm from the truth) -- and
    so are two runs of the SAME HIP code on these draws (tools/replay_twice.py, profiles/r06_replay_twice.txt: bit-identical for 34
    iterations, then 1.5 mm apart at frame 4 and 5 - 11 mm at frames 5 - 13): amplification, not a parity defect.
    Held: every draw, the loop's decisions, the first iterations' losses; per-frame positions within 4 cm of the reference's estimate and the
    ATE within 25 % (the deterministic per-frame comparison from the reference's own state is test_g16_policy_one_frame_at_a_time_...).
    """
    import unislam_amd as us
    from g15_settings import G15
    g = golden("g15_sequence")
    slam, dev = _replay_against(us, g, G15)
    err = (slam.estimate_c2w_list[:, :3, 3] - slam.gt_c2w_list[:, :3, 3]).norm(dim=-1)
    ate, ref_ate = slam.ate_rmse(), float(g["ate_rmse_m"])
    print("g15 replay: |t - t_ref| per frame (mm):", np.array2string(1e3 * dev.numpy(), precision=3, max_line_width=200))
    print(f"g15 replay: ATE {100 * ate:.3f} cm (reference {100 * ref_ate:.3f}), max {100 * float(err.max()):.2f} cm")
    assert slam.mapper.keyframe_list == [int(k) for k in g["keyframe_list"]]
    assert [slam.history["track_iters"].get(i, 0) for i in range(len(g["track_iters"]))] == [int(k) for k in g["track_iters"]]
    assert [m["iters"] for m in slam.history["mapped"]] == [int(k) for k in g["map_iters"]]
    mine, theirs = np.array(slam.history["losses"]), g["loss_log"].astype(np.float64)
    rel = np.abs(mine - theirs) / np.abs(theirs)
    assert rel[0] < 1e-5 and rel[:4].max() < 1e-4 and rel[:8].max() < 5e-3, rel[:8]
    assert float(dev.max()) < 0.04 and abs(ate - ref_ate) <= 0.25 * ref_ate, (float(dev.max()), ate, ref_ate)


def _resume(us, g, P, k):
    """the HIP drivers with the loop's state as the REFERENCE had it at the start of frame k (fixture g16's snapshot): tables, decoders,
    estimated poses, keyframes (pools rebuilt from the stored pixel indices), keyframe poses, iteration counts, flags, and the generator"""
    tag = f"snap{k}__"
    draws = _CheckedDraws(g)
    draws.src.g.set_state(torch.from_numpy(g[tag + "rng_state"]))
    draws.k = int(g[tag + "draw_pos"])
    slam, frames = _g15_slam(us, g, P=P, draws=draws)
    with torch.no_grad():
        slam.es.params.copy_(torch.from_numpy(g[tag + "table_sdf"]).to(DEV)); slam.ec.params.copy_(torch.from_numpy(g[tag + "table_color"]).to(DEV))
    slam.decoders.load_state_dict({key[len(tag + "dec__"):].replace("__", "."): torch.from_numpy(v) for key, v in g.items() if key.startswith(tag + "dec__")})
    slam.estimate_c2w_list[:k] = torch.from_numpy(g["est_c2w"][:k]).to(DEV)
    slam.gt_c2w_list[:k] = torch.from_numpy(g["gt_c2w"][:k]).to(DEV)
    m = slam.mapper
    m.init_phase = False
    for j in range(int(g[tag + "n_keyframes"])):
        kf = int(g["keyframe_list"][j])
        _, color, depth, gt_c2w, rays_d = frames[kf]
        ind = torch.from_numpy(g["kf_pool_idx"][j].astype(np.int64)).to(DEV)
        row = m.arena.alloc()
        m.arena.put(row, color.reshape(-1, 3)[ind], depth.reshape(-1)[ind], rays_d.reshape(-1, 3)[ind])
        m.kf_c2w[row] = torch.from_numpy(g[tag + "kf_est_c2w"][j]).to(DEV)
        m.keyframe_list.append(kf)
        m.keyframe_dict.append({"gt_c2w": gt_c2w, "idx": kf, "row": row, "has_zero": False})
    slam.tracker.num_cam_iters = int(g[tag + "num_cam_iters"])
    slam.m_iters, slam.tracking_back = int(g[tag + "m_iters"]), bool(int(g[tag + "tracking_back"]))
    return slam, draws


def test_g16_policy_one_frame_at_a_time_from_the_reference_state(golden):
    """
    The loop's POLICY against the reference's, deterministically (r6).  Fixture g16 is the reference's loop with a mapped frame every 3rd
    frame, a keyframe every 2nd of those, ACTIVATED MAPPING on with a threshold the run crosses (14 tracking-back frames: doubled tracking /
    mapping iterations, frames mapped and kept as keyframes out of turn, src/Tracker.py:352-363, src/Mapper.py:487,514) and a mapping window
    of 4, so that keyframe_selection_LC's tracking-back branch picks the 3 best-overlapping keyframes (src/Mapper.py:253-272).
    A whole-sequence replay (test above) shows per-iteration losses that agree to 1e-6 at the start and drift apart by a factor ~2 per Adam
    step of the first mapped frame (lr 0.25 on tables initialised at 1e-4: Adam's first steps are +-lr * sign(g), and an entry whose gradient
    nearly cancels flips its sign on a 1e-7 difference) -- any two implementations of the reference do that, so whole-sequence poses and
    threshold decisions can only be compared statistically.  g16 therefore stores the reference loop's WHOLE state at the start of four
    frames, and each is replayed for ONE frame of the loop -- tracking, the uncertainty decision, keyframe selection, the mapped frame's
    iterations, the new keyframe -- from the reference's own tables, decoders, poses, keyframes and generator state:
      frame 12  the count doubles at iteration 9 and falls back at 19; mapped (8 iterations), joint_opt over the whole list, new keyframe
      frame 16  tracking back: 20 + 16 iterations, window = the 3 best-overlapping keyframes + the last two + the frame; kept as a keyframe
      frame 18  starts doubled, falls back; mapped: a 15-frame joint_opt window; new keyframe
      frame 22  tracked only (10 iterations)
    Held per frame: every draw, every decision (iteration counts, flag, mapped or not, window, keyframe list), the iterations' losses (the
    first three 1e-4, nine in ten 5e-3: a ray that changes its loss mask moves one loss by percents), the tracked pose 2e-4 m (measured
    1e-6 .. 5e-5), the poses joint_opt wrote back 1e-3 m (measured 5e-5 .. 1.5e-4).
    """
    import unislam_amd as us
    from g15_settings import G16
    g = golden("g16_policy")
    assert int(g["tracking_back"].sum()) >= 10 and len(set(int(k) for k in g["track_iters"])) >= 3             # the fixture crosses the policy
    for k in G16["snapshots"]:
        slam, draws = _resume(us, g, G16, k)
        slam.run(n_frames=k + 1, start=k, total=G16["n_frames"])
        tag = f"snap{k}__"
        n_draws = int(np.sum(g["draw_frame"] == k))                       # draws: exactly the reference's for this frame
        assert draws.k == int(g[tag + "draw_pos"]) + n_draws, (k, draws.k, int(g[tag + "draw_pos"]), n_draws)
        assert slam.history["track_iters"][k] == int(g["track_iters"][k]) and int(slam.history["tracking_back"][k]) == int(g["tracking_back"][k]), k
        mapped = k in [int(x) for x in g["mapped_frames"]]
        assert len(slam.history["mapped"]) == int(mapped)
        l0 = int(g[tag + "loss_pos"])
        mine = np.array(slam.history["losses"])
        assert int(np.sum(g["loss_frame"] == k)) == len(mine)
        theirs = g["loss_log"][l0:l0 + len(mine)].astype(np.float64)
        n_t = int(g["track_iters"][k])
        rel = np.abs(mine - theirs) / np.abs(theirs)
        est, ref = slam.estimate_c2w_list[k].cpu().numpy(), g["est_c2w"][k]
        print("    mapping losses", mine[n_t:][:4], theirs[n_t:][:4])
        print(f"g16 frame {k}: {n_t} tracking + {len(mine) - n_t} mapping iterations; loss rel diff tracking max {rel[:n_t].max():.1e}, mapping "
              f"{np.array2string(rel[n_t:], precision=1, max_line_width=250)}; |t - t_ref| {np.abs(est[:3, 3] - ref[:3, 3]).max():.1e} m, "
              f"|R - R_ref| {np.abs(est[:3, :3] - ref[:3, :3]).max():.1e}")
        # (one ray crossing the tracker's 10 x median gate -- by construction the highest-loss rays -- moves ONE iteration's loss by percents:
        #  measured 4.9e-2 at one iteration of frame 18, whose final pose then agrees to 1e-6; so: nine in ten within 5e-3, all within 0.1)
        assert rel[:3].max() < 1e-4 and np.sort(rel)[int(0.9 * len(rel))] < 5e-3 and rel.max() < 0.1, (k, rel)
        if mapped:
            j = int(g[tag + "n_mapped"])
            assert slam.history["mapped"][0]["iters"] == int(g["map_iters"][j]) and slam.history["mapped"][0]["joint"] == bool(g["joint_opt"][j])
            off, flat = g["selected_off"], g["selected_flat"]
            nk = int(g[tag + "n_keyframes"])
            # the window = what keyframe_selection_LC returned + the last two keyframes + the current frame (src/Mapper.py:306-310)
            want = sorted([int(x) for x in flat[off[j]:off[j + 1]]] + ([nk - 1, nk - 2] if nk > 1 else [])) + [-1]
            assert slam.history["mapped"][0]["frames"] == want, (k, slam.history["mapped"][0]["frames"], want)
            after = g[tag + "kf_est_c2w_after"]
            assert slam.mapper.keyframe_list == [int(x) for x in g["keyframe_list"][:len(after)]]
            mine_kf = torch.stack([slam.mapper.keyframe_pose(i) for i in range(len(after))]).cpu().numpy()
            print(f"    keyframe poses after the window: max |dt| {np.abs(mine_kf[:, :3, 3] - after[:, :3, 3]).max():.1e} m")
            np.testing.assert_allclose(mine_kf[:, :3, 3], after[:, :3, 3], atol=1e-3)
        # the frame's pose: tracked (and, if mapped with joint_opt, refined by the window)
        np.testing.assert_allclose(est[:3, 3], ref[:3, 3], atol=2e-4 if not mapped else 1e-3)
        np.testing.assert_allclose(est[:3, :3], ref[:3, :3], atol=4e-4 if not mapped else 1e-3)


@pytest.mark.parametrize("k,nf", [(12, 3), (16, 2)])
def test_g16_consecutive_frames_from_a_snapshot_carry_the_loop_state(golden, k, nf):
    """the hand-over BETWEEN frames, from the reference's state at frame k (fixture g16) over nf consecutive frames: the new keyframe and its
    pool, the poses joint_opt wrote back, the iteration counts and the tracking-back flag the tracker leaves for the mapper and for its own
    next frame, the constant-speed prediction from the refined pose.  k = 12: a mapped frame (joint_opt, new keyframe) followed by two
    tracked-only ones whose counts double and fall back; k = 16: two tracking-back frames in a row (both mapped out of turn with 16
    iterations, both kept as keyframes).  The later frames start from a map that is 8 - 32 Adam steps away from the reference's, so their
    bars are those of a short amplification: draws and decisions exact, poses 2e-3 m (measured: 4e-5, 2e-4, 1e-4 | 5e-5, 2.4e-4).  The
    frames are chosen where every uncertainty evaluation is > 20 % from the threshold; measured further out (tools/try_g16_run.py): from
    frame 12 the decisions of nine frames and poses within 1e-3 for six (then 2 - 3e-3); from frame 16 the run leaves the reference's
    decisions at frame 20 (an evaluation 30 % above the threshold there): the amplification, at the trained map's slower rate."""
    import unislam_amd as us
    from g15_settings import G16
    g = golden("g16_policy")
    slam, draws = _resume(us, g, G16, k)
    slam.run(n_frames=k + nf, start=k, total=G16["n_frames"])
    frames = list(range(k, k + nf))
    n_draws = int(np.sum(np.isin(g["draw_frame"], frames)))
    assert draws.k == int(g[f"snap{k}__draw_pos"]) + n_draws, (draws.k, int(g[f"snap{k}__draw_pos"]), n_draws)
    for f in frames:
        assert slam.history["track_iters"][f] == int(g["track_iters"][f]) and int(slam.history["tracking_back"][f]) == int(g["tracking_back"][f]), f
    mapped_ref = [int(x) for x in g["mapped_frames"] if int(x) in frames]
    assert [m["idx"] for m in slam.history["mapped"]] == mapped_ref
    j0 = int(g[f"snap{k}__n_mapped"])
    assert [m["iters"] for m in slam.history["mapped"]] == [int(x) for x in g["map_iters"][j0:j0 + len(mapped_ref)]]
    n_kf = int(g[f"snap{k}__n_keyframes"]) + sum(1 for f in mapped_ref if f in [int(x) for x in g["keyframe_list"]])
    assert slam.mapper.keyframe_list == [int(x) for x in g["keyframe_list"][:n_kf]]
    for f in frames:
        est, ref = slam.estimate_c2w_list[f].cpu().numpy(), g["est_c2w"][f]
        print(f"g16 from {k}: frame {f}: |t - t_ref| {np.abs(est[:3, 3] - ref[:3, 3]).max():.1e} m, |R - R_ref| {np.abs(est[:3, :3] - ref[:3, :3]).max():.1e}")
        np.testing.assert_allclose(est[:3, 3], ref[:3, 3], atol=2e-3)
        np.testing.assert_allclose(est[:3, :3], ref[:3, :3], atol=2e-3)


def test_soak_on_the_closed_loop_bounds_every_frame():
    """
    264 frames (1.2 rounds of SyntheticRoom(path="loop"): the camera comes back to where it started) at the default policy -- a mapped frame
    and a keyframe every 4th frame: 66+ keyframes, so most windows are of the > 20-keyframe kind with the extra rays of the newest frames
    (src/Mapper.py:385-393) and the keyframe arena is walked far beyond a window.  Bounds the LARGEST per-frame error, not only the RMSE, and
    the error of the last round against the first (no growth).  (r4's 300-frame run on the default arc drifted by 8.6 cm from frame 200 on:
    that arc leaves the room through a wall at frame 194 -- SyntheticRoom now refuses such a path.)
    """
    import unislam_amd as us
    from unislam_amd.synthetic import SyntheticRoom
    with pytest.raises(ValueError, match="frame 191"):
        SyntheticRoom(n_frames=300, H=12, W=16, device=DEV)                  # the arc r4's soak ran on
    n = 264
    slam, frames = _build(us, n, mlp_precision="bf16", room=dict(path="loop", tex_freq=4.0), every=4)
    slam.run()
    err = (slam.estimate_c2w_list[:n, :3, 3] - slam.gt_c2w_list[:n, :3, 3]).norm(dim=-1)
    kc = slam.mapper.kind_counts
    print(f"soak: ATE {100 * slam.ate_rmse():.2f} cm, max {100 * float(err.max()):.2f} cm at frame {int(err.argmax())}, last 40 frames max "
          f"{100 * float(err[-40:].max()):.2f} cm, keyframes {len(slam.mapper.keyframe_list)}, LC {slam.mapper.LC_cnt}, kinds {kc}")
    assert len(slam.mapper.keyframe_list) > 60
    assert sum(v for k, v in kc.items() if k[1]) >= 30                       # windows with the extra rays
    assert float(err.max()) < 0.07, float(err.max())                         # every frame within 7 cm (measured over rounds 5-6: 3.5 - 4.7 cm, at the bare wall of frames 35 - 51)
    assert slam.ate_rmse() < 0.03
    assert float(err[-40:].max()) < 0.04                                     # the second pass over the start is no worse than the first
    rot = torch.linalg.matrix_norm(slam.estimate_c2w_list[:n, :3, :3] - slam.gt_c2w_list[:n, :3, :3])
    assert float(rot.max()) < 0.08, float(rot.max())


def test_slam_recovers_the_trajectory():
    import unislam_amd as us
    n = 26
    slam, frames = _build(us, n, graph_replay=False)                       # eager iterations; the default (replayed graphs) below
    est = slam.run()
    print('ATE', slam.ate_rmse(), 'LC', slam.mapper.LC_cnt, 'kf', slam.mapper.keyframe_list)
    gt = slam.gt_c2w_list[:n]
    ate = slam.ate_rmse()
    # dead reckoning from frame 1 on (constant velocity with the first two true poses) drifts much further
    travelled = float((gt[1:, :3, 3] - gt[:-1, :3, 3]).norm(dim=-1).sum())
    assert travelled > 0.4
    assert ate < 0.02, ate                                                  # 2 cm over ~0.5 m of motion
    _, res = slam.evaluate()                                                # the reference's report: after Horn alignment, in cm
    print(res)
    assert res['compared_pose_pairs'] == n and res['error.rmse'] <= 100 * ate + 0.01
    rot_err = torch.linalg.matrix_norm(est[:, :3, :3] - gt[:, :3, :3]).max()
    assert float(rot_err) < 0.05
    m = slam.mapper
    assert len(m.keyframe_list) >= n // 2 - 1 and m.keyframe_list[0] == 0
    assert m.joint_opt                                                      # > 4 keyframes: window poses optimised jointly
    # the map explains the last frame: render it at the ESTIMATED pose and compare with the measured depth
    rend = us.Renderer({"rendering": {"perturb": False, "n_stratified": 32, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid"},
                       type("U", (), dict(bound=slam.bound, device=DEV, H=frames.H, W=frames.W, fx=frames.fx, fy=frames.fy,
                                          cx=frames.cx, cy=frames.cy))())
    _, color, depth, _, _ = frames[n - 1]
    with torch.no_grad():
        out = rend.render_img(([slam.es], [slam.ec]), slam.decoders, est[n - 1], 0.06, DEV, gt_depth=depth)
    d_hat = out[0]
    err = (d_hat.float().reshape(-1) - depth.reshape(-1)).abs()
    assert float(err.median()) < 0.03, float(err.median())


def test_slam_with_replayed_mapping_windows():
    """the DEFAULT settings: every mapped frame binds its window (arena rows, poses, shape) to one of a handful of graphs captured ahead
    (slam.Mapper.prewarm: poses fixed / joint_opt, keyed by the kind of window, not by its number of frames) and replays it `iters`
    times; the trajectory comes out as with eager iterations, and no graph is captured inside the sequence"""
    import unislam_amd as us
    n = 16
    slam, frames = _build(us, n, mlp_precision="bf16")
    assert slam.cfg["mapping"].get("graph_replay", True) is True          # (an extension key: on unless a config turns it off)
    slam.run()
    assert slam.mapper.joint_opt and slam.ate_rmse() < 0.02, slam.ate_rmse()
    kinds = sorted(slam.mapper._wins)
    assert all(w._graph is not None for w in slam.mapper._wins.values())
    # first frame (lr x 5), poses fixed, joint_opt, joint_opt + extra rays (captured ahead, unused in a 16-frame run): windows of 1 .. 8 frames
    assert len(kinds) == 4 and len(slam.mapper.keyframe_list) >= 7, kinds


def test_slam_survives_a_growing_keyframe_arena():
    """an arena sized for 2 keyframes in a run that makes 6: KeyframeArena.grow() doubles the store (new addresses), the mapper drops
    the graphs that hold the old ones and captures again -- the trajectory is as good as with a big enough arena"""
    import unislam_amd as us
    from unislam_amd.synthetic import SyntheticRoom
    from unislam_amd.slam import SLAM
    n = 12
    slam, frames = _build(us, n, mlp_precision="bf16")
    slam.cfg["mapping"]["arena_keyframes"] = 2
    slam.mapper = type(slam.mapper)(slam)                                    # (the arena is sized when the mapper is built)
    assert slam.mapper.arena.K == 3
    slam.run()
    assert slam.mapper.arena.generation >= 1 and slam.mapper.arena.K >= 6 and len(slam.mapper.keyframe_list) >= 5
    assert slam.mapper.kf_c2w.shape[0] == slam.mapper.arena.K
    assert slam.ate_rmse() < 0.02, slam.ate_rmse()


def test_arena_window_with_extra_rays_and_depth_holes():
    """ArenaWindow with the extra block of the newest frames (src/Mapper.py:385-393) and pools that hold pixels without a depth, against
    MapWindow on the same frames, pixel indices, jitter and zero-depth draws: one joint_opt iteration, replayed from the graph"""
    import unislam_amd as us
    from test_gpu_window import _window, _cfg, _ecfg, BOUND, W, LR
    P, rows_a, rows_b = 300, 300, 80
    g = torch.Generator().manual_seed(9)
    arena = us.KeyframeArena(8, P, DEV)
    c2ws, depths, colors, dirs = _window(6, P, 23)
    depths[:, ::9] = 0.0
    for k in range(6):
        arena.put(arena.alloc(), colors[k].to(DEV), depths[k].to(DEV), dirs[k].to(DEV))

    def scene():
        torch.manual_seed(0)
        dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        return us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=rows_a + rows_b)

    step_a, step_p = scene(), scene()
    awin = us.ArenaWindow(step_a, arena, rows_a, rows_b, joint_opt=True, cam_lr=1e-3, has_zero_depth=None)
    awin.capture(t_rand=True, device_draw=False)
    b, n_per, extra = 5, 60, (4, 20)
    frames = [0, 1, 2, 4, 5]
    sel = torch.tensor(frames)
    pwin = us.MapWindow(step_p, c2ws[sel], depths[sel], colors[sel], dirs[sel], n_per, joint_opt=True, cam_lr=1e-3, extra=extra, has_zero_depth=None)
    awin.bind([f + 1 for f in frames], c2ws[sel].to(DEV), n_per, extra)
    R = b * n_per + extra[0] * extra[1]
    idx, idx2 = torch.randint(P, (b, n_per), generator=g), torch.randint(P, extra, generator=g)
    tr = torch.rand(R, 40, generator=g)
    zd = (torch.rand(rows_a + rows_b, 32, generator=g).to(DEV), torch.rand(rows_a + rows_b, 8, generator=g).to(DEV))
    la = pwin.iterate(idx.to(DEV), idx2.to(DEV), t_rand=tr.to(DEV), zero_depth_draws=zd)
    # the arena window's fixed layout: rows [0, rows_a) for the frames' shares, [rows_a, rows_a + rows_b) for the extra block
    awin.t_rand.zero_(); awin.t_rand[:b * n_per].copy_(tr[:b * n_per].to(DEV)); awin.t_rand[rows_a:rows_a + extra[0] * extra[1]].copy_(tr[b * n_per:].to(DEV))
    # (the zero-depth draws are indexed by the compacted row: both windows meet their zero-depth rays in the same order)
    awin.zd_draws[0].copy_(zd[0]); awin.zd_draws[1].copy_(zd[1])
    lb = awin.replay(idx.to(DEV), idx2.to(DEV))
    assert int(step_a.zd_count) == int(step_p.zd_count) > 5
    np.testing.assert_allclose(float(lb), float(la), rtol=1e-5)
    assert torch.allclose(step_a.flat, step_p.flat, rtol=1e-5, atol=1e-6)
    assert torch.allclose(awin.c2ws(), pwin.c2ws(), rtol=0, atol=1e-6)


def test_arena_window_equals_a_plain_window():
    """ArenaWindow (window shape on the device, pools in a KeyframeArena, padded rows) against MapWindow on the same frames, pixel indices
    and jitter: windows of 3 and of 7 frames through ONE captured graph, 4 joint_opt iterations each -- losses, model and poses agree;
    rows beyond b * n_per are flagged invalid"""
    import unislam_amd as us
    from test_gpu_window import _window, _cfg, _ecfg, BOUND, W, LR
    P, rows_a = 400, 420
    g = torch.Generator().manual_seed(4)
    arena = us.KeyframeArena(12, P, DEV)
    c2ws, depths, colors, dirs = _window(9, P, 17)
    for k in range(9):
        row = arena.alloc()
        arena.put(row, colors[k].to(DEV), depths[k].to(DEV), dirs[k].to(DEV))

    def scene():
        torch.manual_seed(0)
        dec = us.Decoders(dict(_cfg(False), model={"mlp_precision": "bf16"}), c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(DEV)
        es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
        with torch.no_grad():
            es.params.copy_(torch.randn(es.params.shape) * 0.3); ec.params.copy_(torch.randn(ec.params.shape) * 0.3)
        return us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=rows_a)

    step_a = scene()
    awin = us.ArenaWindow(step_a, arena, rows_a, 0, joint_opt=True, cam_lr=1e-3, has_zero_depth=False)
    awin.capture(t_rand=True, device_draw=False)
    for frames in ([2, 5, 7], [0, 1, 3, 4, 6, 7, 8]):
        b = len(frames)
        n_per = rows_a // b
        step_p = scene()
        step_a.flat.copy_(step_p.flat); step_a.reset_optimizer(1.0); step_p.reset_optimizer(1.0)
        sel = torch.tensor(frames)
        pwin = us.MapWindow(step_p, c2ws[sel], depths[sel], colors[sel], dirs[sel], n_per, joint_opt=True, cam_lr=1e-3, has_zero_depth=False)
        awin.bind([f + 1 for f in frames], c2ws[sel].to(DEV), n_per)                 # (arena row 0 is the scratch row)
        for it in range(4):
            idx = torch.randint(P, (b, n_per), generator=g)
            tr = torch.rand(b * n_per, 40, generator=g)
            la = pwin.iterate(idx.to(DEV), t_rand=tr.to(DEV))
            awin.t_rand.zero_(); awin.t_rand[:b * n_per].copy_(tr.to(DEV))
            lb = awin.replay(idx.to(DEV))
            np.testing.assert_allclose(float(lb), float(la), rtol=1e-5)
        assert int(step_a.valid[:b * n_per].sum()) == int(step_p.valid[:b * n_per].sum()) and int(step_a.valid[b * n_per:rows_a].sum()) == 0
        assert torch.allclose(step_a.flat, step_p.flat, rtol=1e-5, atol=1e-6)
        assert torch.allclose(awin.c2ws(), pwin.c2ws(), rtol=0, atol=1e-6)
        assert float((awin.c2ws()[1:] - c2ws[sel][1:].to(DEV)).abs().max()) > 1e-4     # the poses did move


def test_keyframe_selection_on_device():
    import unislam_amd as us
    from unislam_amd.slam import keyframe_selection_LC
    from unislam_amd.synthetic import SyntheticRoom
    torch.manual_seed(1)
    frames = SyntheticRoom(n_frames=4, H=120, W=160, device=DEV)
    _, color, depth, c2w, _ = frames[0]
    cam = (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy)
    # keyframes: the same view, a slightly shifted one, one looking the opposite way (+ two trailing entries, always excluded)
    same, shifted, away = c2w.clone(), c2w.clone(), c2w.clone()
    shifted[:3, 3] += torch.tensor([0.05, 0.02, 0.0], device=DEV)
    away[:3, 0] *= -1; away[:3, 2] *= -1
    est = torch.stack([same, shifted, away, same, same])
    sel, pct, loop = keyframe_selection_LC(3, 10, color, depth, c2w, 2, [0, 1, 2, 3, 4], est, cam, DEV)
    # the same camera sees every sample at its own pixel: inside unless within the 20-pixel border, (120*80)/(160*120) = 0.5
    assert 0.35 < float(pct[0]) < 0.65 and float(pct[1]) > 0.25 and float(pct[2]) == 0.0 and not loop
    assert sel == [0, 1, 2]                                                 # "global": every keyframe joins the window
    sel, pct, loop = keyframe_selection_LC(3, 10, color, depth, c2w, 2, [0, 1, 2, 3, 4], est, cam, DEV, tracking_back=True)
    assert sorted(sel) == [0, 1]                                            # tracking back: the best-overlapping ones
    # device tensors take the fused path (us_keyframe_overlap: one launch for all keyframes); the torch chain of src/Mapper.py:188-240 on
    # the SAME pixel draw gives the same shares (a point within rounding of an image border may count differently: 1 / 400)
    from unislam_amd import slam as S
    from unislam_amd import _lib as L
    g = torch.Generator(device=DEV).manual_seed(3)
    pix = torch.randint(frames.H * frames.W, (50,), device=DEV, generator=g)
    d = depth.clone(); d.view(-1)[pix[:5]] = 0.0                            # pixels without a depth are left out
    out = torch.zeros(3, device=DEV)
    kf = torch.tensor([0, 1, 2], device=DEV)
    L.check(L.lib().us_keyframe_overlap(L.ptr(c2w.contiguous()), L.ptr(d), L.ptr(pix), 50, 8, L.host_floats([frames.fx, frames.fy, frames.cx, frames.cy]),
                                        frames.H, frames.W, 20, L.ptr(est.contiguous()), L.ptr(kf), 3, L.ptr(out), L.stream()), "us_keyframe_overlap")
    x, y = (pix % frames.W).float(), (pix // frames.W).float()
    dirs = torch.stack([(x - frames.cx) / frames.fx, -(y - frames.cy) / frames.fy, -torch.ones_like(x)], -1)
    rd = (dirs[:, None, :] * c2w[:3, :3]).sum(-1); ro = c2w[:3, 3].expand(rd.shape)
    gd = d.view(-1)[pix]
    nz = gd > 0
    t = torch.linspace(0., 1., 8, device=DEV)
    z = gd[nz][:, None] * 0.8 * (1 - t) + (gd[nz][:, None] + 0.5) * t
    pts = (ro[nz][:, None] + rd[nz][:, None] * z[..., None]).reshape(-1, 3)
    ref = S.keyframe_overlap(pts, est[:3], frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy)
    assert float((out - ref).abs().max()) <= 1.01 / pts.shape[0], (out, ref)
    assert float(out[0]) > 0.3 and float(out[2]) == 0.0


def test_slam_from_a_sequence_on_disk(tmp_path):
    """the configured system end to end: a YAML in the reference's key layout + a Replica-layout sequence on disk (JPEG colour, 16-bit
    PNG depth, traj.txt) written by export_sequence -> config.build_slam (bound, resolutions, encoders, decoders, reader) -> run"""
    import yaml
    import unislam_amd as us
    from unislam_amd import config as C, datasets as D
    from unislam_amd.synthetic import SyntheticRoom
    torch.manual_seed(0)
    n = 12
    frames = SyntheticRoom(n_frames=n, H=120, W=160, device=DEV)
    folder = D.export_sequence(frames, str(tmp_path / "room"), layout="replica", png_depth_scale=6553.5)
    w = lambda fs, c, t, d, col: {"w_sdf_fs": fs, "w_sdf_center": c, "w_sdf_tail": t, "w_depth": d, "w_color": col}
    cfg = {"dataset": "replica", "scale": 1, "device": DEV, "m_mask_mode": "original", "t_mask_mode": "original", "grid_mode": "hash_grid",
           "planes_res": {"bound_dividable": 0.24},
           "grid": {"enc": "HashGrid", "hash_size_sdf": 16, "hash_size_color": 16, "voxel_sdf": 0.02, "voxel_color": 0.02, "tcnn_network": False},
           "tracking": dict(ignore_edge_W=8, ignore_edge_H=8, const_speed_assumption=True, lr_T=0.002, lr_R=0.001, pixels=1000, iters=10,
                            activated_mapping_mode=True, uncertainty_ts=0.001, **w(10, 200, 50, 1, 5)),
           "mapping": dict(every_frame=2, keyframe_every=2, joint_opt=True, joint_opt_cam_lr=0.001, mapping_window_size=20, lr_first_factor=5,
                           lr_factor=1, pixels=2000, iters_first=300, iters=20, LC=True, LC_ts=0.95, bound=[[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]],
                           lr={"decoders_lr": 0.001, "hash_grids_lr": 0.05, "c_hash_grids_lr": 0.05}, **w(5, 200, 10, 0.1, 5)),
           "cam": dict(H=frames.H, W=frames.W, fx=frames.fx, fy=frames.fy, cx=frames.cx, cy=frames.cy, png_depth_scale=6553.5, crop_edge=0),
           "rendering": {"n_stratified": 32, "n_importance": 8, "perturb": True, "learnable_beta": True},
           "model": {"c_dim": 32, "truncation": 0.06}, "data": {"input_folder": folder}}
    path = tmp_path / "room.yaml"
    path.write_text(yaml.safe_dump(cfg))
    slam = C.build_slam(C.load_config(str(path)))
    assert len(slam.frames) == n and slam.frames[0][1].device.type == "cpu"
    assert tuple(slam.bound[:, 1].tolist()) == pytest.approx((6.7, 3.7, 1.66), abs=1e-5)          # enlarged to multiples of 0.24
    slam.run()
    ate = slam.ate_rmse()
    _, res = slam.evaluate()
    print("ATE from disk", ate, res)
    assert ate < 0.02, ate


def test_config5_standin_full_loop_4096_rays_bf16():
    """
    BASELINE configs[4] asks for TUM fr1_desk, the full tracking + mapping loop with 4096 mapping rays on the bf16 MFMA path, judged by
    ATE.  The TUM sequence is NOT in this container (no datasets, no network), so this is a stand-in: the same loop (Tracker + Mapper
    drivers, src/Tracker.py:271-370, src/Mapper.py:461-545) on the synthetic orbit with the TUM hyper-parameters that shape the hot path
    -- 4096 mapping pixels, 48 + 8 samples, fixed beta (tum.yaml:49), tables 16 / 16, table lr 0.02, mlp_precision bf16 -- and the
    reference's reports: ATE (eval_ate.py) and the render quality of the final map (eval_recon.py:235-307: PSNR, depth L1).
    """
    import unislam_amd as us
    from unislam_amd.synthetic import SyntheticRoom
    from unislam_amd.slam import SLAM
    torch.manual_seed(0)
    n = 21
    frames = SyntheticRoom(n_frames=n, H=120, W=160, device=DEV)
    bound = O.load_bound([[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]])
    res = int((bound[:, 1] - bound[:, 0]).max() / 0.02)
    ecfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 16, "base_resolution": 16,
            "per_level_scale": O.per_level_scale(res)}
    es, ec = us.HashGridEncoding(3, ecfg).to(DEV), us.HashGridEncoding(3, ecfg).to(DEV)
    cfg = {"rendering": {"perturb": True, "n_stratified": 48, "n_importance": 8}, "scale": 1, "grid_mode": "hash_grid",
           "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06, learnable_beta=False).to(DEV)
    dec.bound = bound
    slam = SLAM(frames, (frames.H, frames.W, frames.fx, frames.fy, frames.cx, frames.cy), es, ec, dec, bound,
                cfg={"tracking": dict(ignore_edge_W=8, ignore_edge_H=8, pixels=2000, iters=10),
                     "mapping": dict(pixels=4096, iters=20, iters_first=300, every_frame=2, keyframe_every=2,
                                     lr=dict(decoders=0.001, sdf_grid=0.02, color_grid=0.02)),
                     "rendering": dict(n_stratified=48, n_importance=8, perturb=True)})
    assert slam.mapper.step.desc_s.precision == 1 and slam.mapper.step.S == 56          # bf16 MFMA decoders, 48 + 8 samples
    slam.run()
    _, rep = slam.evaluate()
    ate = slam.ate_rmse()
    rq = slam.evaluate_rendering(stride=5)
    print("config-5 stand-in: ATE rmse [cm, Horn-aligned]", rep["error.rmse"], "unaligned [m]", ate, "render", rq)
    assert rep["compared_pose_pairs"] == n
    assert ate < 0.02, ate                                                  # 2 cm over ~0.4 m of motion
    assert rq["frames"] == 5 and rq["avg_psnr"] > 18.0 and rq["depth_l1_render"] < 0.05, rq


@pytest.mark.parametrize("H,W", [(680, 1200), (120, 160), (7, 5)])
def test_pool_cut_is_a_random_subset_without_repetition(H, W):
    """us_pool_cut (KeyframeArena.cut): src/Mapper.py:329-337's `randperm(H * W)[:10 %]` + three gathers as one launch -- the pool's rows are
    DISTINCT pixels of the frame with their colour / depth / direction, a different subset per seed, spread evenly over the frame; the
    flag reports pixels without a depth"""
    import unislam_amd as us
    g = torch.Generator().manual_seed(H)
    n = H * W
    color = torch.rand(H, W, 3, generator=g).to(DEV); dirs = torch.randn(H, W, 3, generator=g).to(DEV)
    depth = (torch.arange(n, dtype=torch.float32).reshape(H, W) + 1.0).to(DEV)      # depth = pixel number + 1: the pool tells which pixels it took
    P = max(1, int(n * 0.1))
    arena = us.KeyframeArena(3, P, DEV)
    flag = arena.cut(1, color, depth, dirs, seed=5)
    pix = (arena.depth[1] - 1.0).long()
    assert int(flag) == 0 and int(pix.min()) >= 0 and int(pix.max()) < n
    assert torch.unique(pix).numel() == P                                           # no pixel twice
    assert torch.equal(arena.color[1], color.reshape(-1, 3)[pix]) and torch.equal(arena.dirs[1], dirs.reshape(-1, 3)[pix])
    arena.cut(2, color, depth, dirs, seed=6)
    if P > 8:
        assert not torch.equal(arena.depth[1], arena.depth[2])                      # another seed, another subset
        both = torch.cat([pix, (arena.depth[2] - 1.0).long()])
        assert torch.unique(both).numel() > 1.7 * P                                 # two independent 10 % subsets share ~1 % of the frame
    if n > 500000:                                                                  # evenly spread: every sixteenth of the frame holds its share (5100 +- 71)
        share = torch.bincount(pix * 16 // n, minlength=16).float() / P
        assert float((share - 1 / 16).abs().max()) < 0.08 / 16, share
    depth2 = depth.clone(); depth2.view(-1)[pix[:1]] = 0.0                          # a hole at a pixel this seed takes
    assert int(arena.cut(1, color, depth2, dirs, seed=5)) == 1
