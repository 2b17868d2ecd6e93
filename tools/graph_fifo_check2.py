"""The sequence tests/test_gpu_window.py's registry test ran under the first r6 release rule ("fifo", forced here), on its own, with a marker
before every step (a host SIGSEGV in hipGraphLaunch leaves no Python exception): two graphs released in ONE collect() behind a living newer
one, then a new capture.   python -X faulthandler tools/graph_fifo_check2.py [pre]     ("pre": capture and drop four windows first)
Runs clean both ways; the same sequence after the other tests of tests/test_gpu_window.py ends in the SIGSEGV (profiles/r06_graph_release_rules.txt)."""
import gc, os, sys
os.environ["US_GRAPH_RELEASE"] = "fifo"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import unislam_amd as us
from unislam_amd import graph
from test_gpu_window import _window, _cfg, _ecfg, BOUND, W, LR

DEV = "cuda:0"


def say(*a):
    print(*a, flush=True)


def make(seed):
    torch.manual_seed(seed)
    dec = us.Decoders(_cfg(False), c_dim=32, truncation=0.06).to(DEV)
    es, ec = us.HashGridEncoding(3, _ecfg(14)).to(DEV), us.HashGridEncoding(3, _ecfg(15)).to(DEV)
    c2ws, depths, colors, dirs = _window(4, 300, seed)
    step = us.MapStep(es, ec, dec, BOUND, 32, 8, 0.06, W, LR, max_rays=4 * 48)
    win = us.MapWindow(step, c2ws, depths, colors, dirs, 48, joint_opt=True, cam_lr=1e-3, has_zero_depth=False)
    win.capture()
    return win


if "pre" in sys.argv:
    for k in range(4):
        w = make(50 + k)
        for _ in range(3):
            w.replay()
        torch.cuda.synchronize()
        del w
    gc.collect()
    say("pre: registry", len(graph._KEEP), "released", graph.collect())
n0 = len(graph._KEEP)
a, b, c = make(1), make(2), make(3)
say("captured a b c; registry", len(graph._KEEP) - n0)
del b; gc.collect(); say("b dropped, released", graph.collect())
for _ in range(5):
    a.replay(); c.replay()
torch.cuda.synchronize(); say("a, c replayed")
del a; gc.collect(); say("a dropped, released", graph.collect())
for _ in range(10):
    c.replay()
torch.cuda.synchronize(); say("c replayed")
d = make(4); say("d captured")
d.replay(); torch.cuda.synchronize(); say("d replayed once")
c.replay(); torch.cuda.synchronize(); say("c replayed after d")
for _ in range(5):
    d.replay(); c.replay()
torch.cuda.synchronize(); say("d, c replayed: ok")
