"""development (r5): the table gradient cut by levels -- the accumulate pass of the coarse levels beside the record pass of the fine ones on a
second stream (us_hashgrid_bwd_joint_part: EXPERIMENTS build, tools/build_experiments.sh) -- against the whole pass, at the bench shape: same gradients, wall time per variant.
   python tools/split_levels.py"""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench as B
B.torch = torch
import unislam_amd as us
from unislam_amd import _lib as L
dev = "cuda:0"
bound = B.load_bound(B.ROOM0_BOUND)
mk = lambda l2: us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": l2,
                                        "base_resolution": 16, "per_level_scale": B.per_level_scale(816)}).to(dev)
torch.manual_seed(0)
dec = us.Decoders({"grid_mode": "hash_grid", "grid": {"tcnn_network": False}, "model": {"mlp_precision": "bf16"}}, c_dim=32, hidden_size=32,
                  truncation=0.06, n_blocks=2).to(dev)
es, ec = mk(16), mk(19)
st = us.MapStep(es, ec, dec, bound, 48, 16, 0.06, B.W, B.LR, max_rays=4096, deterministic=True)
c2ws, pd, pc, pr = B.keyframe_pools(16, bound, 1000, dev)
win = us.MapWindow(st, c2ws, pd, pc, pr, 256, joint_opt=False, has_zero_depth=False)
win.draw(); win._sample(None, True)
st.forward(win.ro, win.rd, win.gd, win.gc, None, False, None, True, True)
st.backward()                                     # leaves counts, scans, records, dL/dfeatures in place
torch.cuda.synchronize()
lib, P = L.lib(), L.ptr
off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)
ds, dc = ctypes.byref(es.desc), ctypes.byref(ec.desc)
N = 4096 * 64
flags = 3 | L.US_GRID_BWD_OVERWRITE | L.US_GRID_BWD_DETERMINISTIC | L.US_GRID_BWD_COUNTED | L.US_GRID_BWD_SCANNED
ga, gb = off(st.grad, st.o_tab_s), off(st.grad, st.o_tab_c)
whole = lambda q: L.check(lib.us_hashgrid_bwd_joint(ds, dc, P(st.pts), P(st.d_feat_s), P(st.d_feat_c), N, ga, gb, flags, P(st.ws), st.ws_bytes, q), "whole")
part = lambda lo, hi, what, q: L.check(lib.us_hashgrid_bwd_joint_part(ds, dc, P(st.pts), P(st.d_feat_s), P(st.d_feat_c), N, ga, gb, flags, P(st.ws),
                                                                       st.ws_bytes, lo, hi, what, q), "part")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
q = lambda s: ctypes.c_void_p(s.cuda_stream)

def serial(cut):
    part(0, cut, 1, q(main)); part(cut, 16, 1, q(main)); part(0, cut, 2, q(main)); part(cut, 16, 2, q(main))

def overlapped(cuts):
    # record pass of part k+1 on the main stream, accumulate pass of part k on the side stream
    edges = [0] + list(cuts) + [16]
    ev_prev = None
    for k in range(len(edges) - 1):
        part(edges[k], edges[k + 1], 1, q(main))
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            part(edges[k], edges[k + 1], 2, q(side))
    main.wait_stream(side)

def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

st.grad.zero_(); whole(q(main)); torch.cuda.synchronize(); ref = st.grad.clone()
for name, fn in (("serial cut 8", lambda: serial(8)), ("overlapped (8,)", lambda: overlapped((8,))), ("overlapped (10,)", lambda: overlapped((10,))),
                 ("overlapped (6, 11)", lambda: overlapped((6, 11))), ("overlapped (5, 9, 13)", lambda: overlapped((5, 9, 13)))):
    st.grad.zero_(); fn(); torch.cuda.synchronize()
    d = (st.grad - ref).abs().max() / ref.abs().max()
    print(f"{name:24s} max rel diff {float(d):.2e}", flush=True)
print(f"whole pass                 {timed(lambda: whole(q(main))):7.1f} us")
for name, fn in (("serial cut 8", lambda: serial(8)), ("overlapped (8,)", lambda: overlapped((8,))), ("overlapped (10,)", lambda: overlapped((10,))),
                 ("overlapped (12,)", lambda: overlapped((12,))), ("overlapped (6, 11)", lambda: overlapped((6, 11))),
                 ("overlapped (9, 13)", lambda: overlapped((9, 13))), ("overlapped (5, 9, 13)", lambda: overlapped((5, 9, 13)))):
    print(f"{name:24s}   {timed(fn):7.1f} us", flush=True)
