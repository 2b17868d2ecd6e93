"""
Host logic that needs no GPU: the product fails LOUDLY without the HIP path (no CPU fallback anywhere), the analytic test scene refuses
camera paths that leave the room, the registry that keeps captured graphs alive, and the reference-shaped optimiser's bookkeeping.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_drop_in_modules_refuse_cpu_tensors():
    import unislam_amd as us
    cfg = {"rendering": {"perturb": True, "n_stratified": 8, "n_importance": 4}, "scale": 1, "grid_mode": "hash_grid", "grid": {"tcnn_network": False}}
    dec = us.Decoders(cfg, c_dim=32, truncation=0.06)
    ecfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 16, "per_level_scale": 1.3}
    es, ec = us.HashGridEncoding(3, ecfg), us.HashGridEncoding(3, ecfg)
    with pytest.raises(us.UniSlamHipError):
        dec(torch.rand(7, 3), ([es], [ec]))                      # CPU tensors: no fallback, an error
    p = torch.nn.Parameter(torch.randn(5))
    opt = us.optim.Adam([{"params": [p], "lr": 0}])
    opt.param_groups[0]["lr"] = 0.1                              # the reference sets the learning rates afterwards (src/Mapper.py:123-126)
    opt.step()                                                   # no gradient yet: nothing to do, no error
    p.grad = torch.ones(5)
    with pytest.raises(us.UniSlamHipError):
        opt.step()
    with pytest.raises(ValueError):
        us.optim.Adam([p], weight_decay=0.1)
    # the nn.Linear parameters' packing order is the kernels' flat layout: weights (last matrix padded to 16 rows), then biases
    lay = dec._pack_layout()
    n_s, n_c = dec._n_packed
    assert n_s == 32 * 16 + 16 * 16 + 16 * 16 + 16 + 16 + 16 and n_c == n_s
    offs = [o for _, o in lay]
    assert offs == sorted(offs) and offs[0] == 0 and lay[6][1] == n_s and len(lay) == 12
    assert [tuple(p_.shape) for p_, _ in lay[:6]] == [(16, 32), (16, 16), (1, 16), (16,), (16,), (1,)]


def test_synthetic_room_refuses_paths_through_walls():
    from unislam_amd.synthetic import SyntheticRoom
    with pytest.raises(ValueError, match="frame 191"):
        SyntheticRoom(n_frames=300, H=6, W=8, device="cpu")      # the default arc leaves the room through y = 3.0 at frame 194
    ok = SyntheticRoom(n_frames=190, H=6, W=8, device="cpu")
    assert float(ok.sdf(ok.poses[:, :3, 3]).min()) >= 0.05
    loop = SyntheticRoom(n_frames=660, H=6, W=8, device="cpu", path="loop", clearance=0.2)       # three rounds of the closed loop
    pos = loop.poses[:, :3, 3]
    assert float(loop.sdf(pos).min()) > 0.2
    assert float((pos[220] - pos[0]).norm()) < 0.03 and float((pos[440] - pos[0]).norm()) < 0.03 and float((pos[110] - pos[0]).norm()) > 1.3
    with pytest.raises(ValueError):
        SyntheticRoom(n_frames=5, H=6, W=8, device="cpu", path="spiral")
    # the default path is what it always was (fixture g15 and the slam tests' bounds rest on it)
    assert torch.allclose(ok.poses[0, :3, 3], torch.tensor([1.3, 1.2, -0.1])) and abs(float(ok.poses[100, 0, 3]) - 3.14) < 0.01


def test_graph_registry_keeps_and_releases(monkeypatch):
    """the registry of captured graphs on a PRIVATE list (the process-global one holds the graphs of earlier GPU tests): nothing is
    destroyed while ANY owner is alive (destroying one multi-branch hipGraph leaves the living ones with dangling streams); when the last
    owner is gone the registry empties at the next capture"""
    import gc
    from unislam_amd import graph
    monkeypatch.setattr(graph, "_KEEP", [])
    monkeypatch.delenv("US_GRAPH_RELEASE", raising=False)

    class Owner:
        pass
    a, b, c = Owner(), Owner(), Owner()
    ga, gb, gc_ = object(), object(), object()
    graph._keep(ga, a); graph._keep(gb, b); graph._keep(gc_, c)
    assert [e[0] for e in graph._KEEP] == [ga, gb, gc_] and graph.collect() == 0
    del b; gc.collect()
    assert graph.collect() == 0 and len(graph._KEEP) == 3
    del a; gc.collect()
    assert graph.collect() == 0 and len(graph._KEEP) == 3          # c's owner lives: a's and b's graphs wait for it
    monkeypatch.setenv("US_GRAPH_RELEASE", "fifo")                 # the refuted rule, kept for reproducing the fault: the front goes
    assert graph.collect() == 2 and [e[0] for e in graph._KEEP] == [gc_]
    monkeypatch.delenv("US_GRAPH_RELEASE")
    d = Owner(); gd = object()
    graph._keep(gd, d)
    del c; gc.collect()
    assert graph.collect() == 0 and len(graph._KEEP) == 2          # d's owner lives
    del d; gc.collect()
    monkeypatch.setenv("US_GRAPH_RELEASE", "never")
    assert graph.collect() == 0 and len(graph._KEEP) == 2
    monkeypatch.delenv("US_GRAPH_RELEASE")
    assert graph.collect() == 2 and graph._KEEP == []              # every owner gone: all of them, oldest first
    sentinel = object()
    graph._keep(sentinel)                                          # no owner: kept until release_all(), and keeps everything behind it
    e = Owner(); graph._keep(object(), e); del e; gc.collect()
    assert graph.collect() == 0 and len(graph._KEEP) == 2
    assert graph.release_all() == 2 and graph._KEEP == []
    monkeypatch.setenv("US_KEEP_GRAPHS", "0")                      # the r4 behaviour, for reproducing the runtime fault
    graph._keep(sentinel)
    assert graph._KEEP == []


def test_torch_draws_is_torch_s_global_stream():
    """slam.TorchDraws(state) hands out what torch's global CPU generator hands the reference from that state: same numbers for the same
    calls in the same order (randint / rand / randperm interleaved as the loop interleaves them)"""
    from unislam_amd.slam import TorchDraws
    torch.manual_seed(77)
    state = torch.get_rng_state().clone()
    want = [torch.randint(2240, (200,)), torch.rand(197, 40), torch.randint(3072, (50,)), torch.randperm(3072)[:307], torch.rand(0, 40),
            torch.randint(307, (400,)), torch.randperm(0), torch.rand(400, 40)]
    d = TorchDraws(state=state.numpy())
    got = [d.randint(2240, 200), d.rand(197, 40), d.randint(3072, 50), d.randperm(3072)[:307], d.rand(0, 40), d.randint(307, 400), d.randperm(0),
           d.rand(400, 40)]
    for a, b in zip(want, got):
        assert a.shape == b.shape and torch.equal(a, b)
    assert not torch.equal(TorchDraws(seed=1).rand(3, 3), TorchDraws(seed=2).rand(3, 3))


def test_jitter_rows_follow_the_reference_s_compaction():
    """slam._jitter_rows: the reference draws rand(R', S) for the rays that passed the pre-filter AND carry a depth (src/Mapper.py:403-406,
    src/utils/Renderer.py:87-101), then -- if any ray has no depth -- rand(n0, n_strat) and rand(n0, n_imp) for those (Renderer.py:117,
    common.py:64); the kernels take t_rand by ROW and the zero-depth draws by COMPACTED ROW of !(depth > 0), pre-filtered or not"""
    from unislam_amd.slam import TorchDraws, _jitter_rows
    valid = torch.tensor([1, 0, 1, 1, 0, 1, 1, 0], dtype=torch.uint8)
    gd = torch.tensor([1.0, 2.0, 0.0, 3.0, 0.0, 0.5, 0.0, 0.0])
    S, ns, ni = 5, 3, 2
    g = torch.Generator().manual_seed(3)
    ref_t, ref_u0, ref_u1 = torch.rand(3, S, generator=g), torch.rand(2, ns, generator=g), torch.rand(2, ni, generator=g)   # rows 0, 3, 5 | rows 2, 6
    t_rand, zd = _jitter_rows(TorchDraws(seed=3), valid, gd, S, ns, ni, True, True)
    assert torch.equal(t_rand[[0, 3, 5]], ref_t) and float(t_rand[[1, 2, 4, 6, 7]].abs().sum()) == 0.0
    # rows without a depth: 2, 4, 6, 7 -> compacted 0, 1, 2, 3; of those 2 and 6 passed the pre-filter: compacted rows 0 and 2
    assert torch.equal(zd[0][[0, 2]], ref_u0) and torch.equal(zd[1][[0, 2]], ref_u1) and float(zd[0][[1, 3]].abs().sum()) == 0.0
    # no jitter (perturb off): nothing is drawn for t_rand, the inverse-transform draws still are
    t2, zd2 = _jitter_rows(TorchDraws(seed=3), valid, gd, S, ns, ni, False, True)
    g2 = torch.Generator().manual_seed(3)
    assert float(t2.abs().sum()) == 0.0 and torch.equal(zd2[1][[0, 2]], torch.rand(2, ni, generator=g2)) and float(zd2[0].abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="without a depth"):
        _jitter_rows(TorchDraws(seed=3), valid, gd, S, ns, ni, True, False)
    t3, zd3 = _jitter_rows(TorchDraws(seed=3), torch.ones(4, dtype=torch.uint8), torch.ones(4), S, ns, ni, True, False)
    assert zd3 is None and torch.equal(t3, torch.rand(4, S, generator=torch.Generator().manual_seed(3)))
