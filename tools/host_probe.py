"""Host-side cost of enqueueing the SLAM-scale calls (224x224, batch 1): wall time of the call itself (asynchronous: it returns
when the last launch is queued) against the GPU time of the same work.  If enqueue ~ GPU time the pipelined replay is bound by the
launching thread, not by the GPU.      python tools/host_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_slam_amd import weights as W                     # noqa: E402
from vista_slam_amd.sta_frontend import STAFrontend        # noqa: E402
from vista_slam_amd.slam_scheduler import regress_views_begin, regress_views_finish    # noqa: E402

dev = "cuda:0"
H = Wd = 224
imgs = torch.from_numpy(W.synth_images(6, H, Wd, seed=43, tag=3)).to(dev)
m = STAFrontend(W.FULL, dev).load_procedural(seed=43)
feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(6)]
torch.cuda.synchronize()


def probe(name, fn, n=30, drain=True):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        fn()
        host += time.perf_counter() - h0
        if drain:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{name:34s} host enqueue {host / n * 1e3:7.3f} ms   wall per call {tot / n * 1e3:7.3f} ms", flush=True)


probe("encode (drained each call)", lambda: m._encode_image(imgs[:1], None, normalize=False))
probe("encode (back to back)", lambda: m._encode_image(imgs[:1], None, normalize=False), drain=False)
pend = [None]


def beg():
    pend[0] = regress_views_begin(m, feats[5], feats[:5], H, Wd)


def fin():
    regress_views_finish(m, pend[0], [True] * 5, 0.0)


def both():
    beg(); fin()


# begin / finish separately: host time of each phase with an empty GPU queue
for _ in range(3):
    both()
torch.cuda.synchronize()
hb = hf = 0.0
n = 30
for _ in range(n):
    t0 = time.perf_counter(); beg(); hb += time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fin(); hf += time.perf_counter() - t0
    torch.cuda.synchronize()
print(f"scheduler k=5: begin host enqueue {hb / n * 1e3:.3f} ms, finish host enqueue {hf / n * 1e3:.3f} ms (GPU idle at call time)", flush=True)
probe("scheduler k=5 begin+finish", both)
