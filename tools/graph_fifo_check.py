"""Is destroying the OLDEST captured multi-branch hipGraph safe while newer ones keep replaying?  (graph.py's first r6 release rule, "fifo",
rested on it.  This script runs clean -- profiles/r06_graph_fifo.txt -- and the rule was still wrong: profiles/r06_graph_release_rules.txt;
the default rule is now "idle", this script forces the old one.)  profiles/r05_hipgraph_destroy_segv.txt is the opposite order: destroying a NEWER graph breaks the older ones.  Three MapWindows, each
with its captured graph (side streams: scans beside the decoders), released oldest first with replays of the survivors in between; then
twenty capture / drop cycles with one long-lived window replaying throughout.
   python tools/graph_fifo_check.py"""
import gc, os, sys
os.environ["US_GRAPH_RELEASE"] = "fifo"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import unislam_amd as us
from unislam_amd import graph
import unislam_oracle as O
from test_gpu_window import _window, _cfg, _ecfg, BOUND, W, LR

DEV = "cuda:0"


def make(seed):
    torch.manual_seed(seed)
    dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
    c2ws, depths, colors, dirs = _window(6, 400, seed)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=6 * 64)
    win = us.MapWindow(step, c2ws, depths, colors, dirs, 64, joint_opt=True, cam_lr=1e-3, has_zero_depth=False)
    win.capture()
    return win


n0 = len(graph._KEEP)
a, b, c = make(1), make(2), make(3)
assert len(graph._KEEP) == n0 + 3
for w in (a, b, c):
    for _ in range(5):
        w.replay()
torch.cuda.synchronize()
del a; gc.collect()
assert graph.collect() == 1, "the oldest graph's owner is gone: it must be released"
for _ in range(20):
    b.replay(); c.replay()
torch.cuda.synchronize()
print("oldest destroyed, the two newer graphs replay: ok")
del b; gc.collect()
assert graph.collect() == 1
for _ in range(20):
    c.replay()
torch.cuda.synchronize()
print("second oldest destroyed, the newest replays: ok")
for k in range(20):                                            # c is now the OLDEST and stays; younger ones come and go behind it
    w = make(10 + k)
    for _ in range(3):
        w.replay(); c.replay()
    del w; gc.collect()
torch.cuda.synchronize()
print(f"20 capture / drop cycles behind a long-lived window: registry holds {len(graph._KEEP) - n0} graphs (they wait for the oldest)")
del c; gc.collect()
n = graph.collect()
print(f"the long-lived window dropped: {n} graphs released oldest-first, registry {len(graph._KEEP) - n0}")
d = make(99)
for _ in range(10):
    d.replay()
torch.cuda.synchronize()
print("a fresh window after the release: ok")
