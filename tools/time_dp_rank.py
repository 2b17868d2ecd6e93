"""development: python tools/time_dp_rank.py -- what ONE rank of the data-parallel step costs (4096 rays x 64 over 16 keyframes, room0 tables, bf16
decoders): a 1-rank RCCL group, so the collectives are issued and waited for but move nothing.  MapWindow on MapStep(group=True): eager,
replayed as hipGraph segments between the collectives, and -- if the runtime captures RCCL calls -- as ONE graph; the single process beside it."""
import os, sys, time, json
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import unislam_amd as us
import bench as B

B.torch = torch
dist.init_process_group("nccl", rank=0, world_size=1)
dev = "cuda:0"
bound = B.load_bound(B.ROOM0_BOUND)
pls = B.per_level_scale(int((bound[:, 1] - bound[:, 0]).max() / 0.01))
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": pls}).to(dev)
c2ws, pd, pc, pr = B.keyframe_pools(16, bound, 3000, dev)


def build(**kw):
    torch.manual_seed(0)
    cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}
    dec = us.Decoders(cfg, c_dim=32, hidden_size=32, truncation=0.06, n_blocks=2).to(dev)
    return us.MapStep(mk(16), mk(19), dec, bound, 48, 16, 0.06, B.W, B.LR, max_rays=4096, **kw)


def timed(fn, k=100):
    for _ in range(10):
        fn()
    rounds = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        rounds.append(round(1e3 * (time.perf_counter() - t) / k, 4))
    return sorted(rounds)[1]


out = {}
for jo in (False, True):
    tag = "joint_opt" if jo else "poses_fixed"
    for name, kw in (("local_fast", dict(group=True)), ("colour_first", dict(group=True, dp_mode="colour_first"))):
        win = us.MapWindow(build(**kw), c2ws, pd, pc, pr, 256, joint_opt=jo, has_zero_depth=False)
        out[f"{tag} {name} eager"] = timed(win.iterate)
        win.capture(collectives="between")
        out[f"{tag} {name} segments ({len(win._graph.segments)})"] = timed(win.replay)
        try:
            win.capture(collectives="inside")
            out[f"{tag} {name} one graph, collectives inside"] = timed(win.replay)
        except Exception as e:
            out[f"{tag} {name} one graph, collectives inside"] = repr(e)[:200]
    win = us.MapWindow(build(), c2ws, pd, pc, pr, 256, joint_opt=jo, has_zero_depth=False)
    out[f"{tag} single process eager"] = timed(win.iterate)
    win.capture()
    out[f"{tag} single process replayed"] = timed(win.replay)
print(json.dumps(out, indent=1))
dist.destroy_process_group()
