"""How much do independent SLAM-scale chains gain from running concurrently on one MI355X?  (224x224, batch 1: every chain is
~200 dependent dispatches that each leave most of the chip idle.)

    python tools/conc_probe.py [iters]

  A  encode alone, scheduler (5 edges, all accepted) alone                         -> ms per call
  B  encode on a second stream enqueued under the scheduler call (one handle)      -> ms per (encode + scheduler)
  C  two handles, two host threads, each running scheduler calls                   -> aggregate scheduler calls / s
  D  C + a third thread running encodes on a third handle
"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_slam_amd import weights as W                     # noqa: E402
from vista_slam_amd.sta_frontend import STAFrontend        # noqa: E402
from vista_slam_amd.slam_scheduler import regress_views    # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
H = Wd = 224
imgs = torch.from_numpy(W.synth_images(6, H, Wd, seed=43, tag=3)).to(dev)


def make():
    return STAFrontend(W.FULL, dev).load_procedural(seed=43)


m = make()
feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(6)]
torch.cuda.synchronize()


def enc(mm):
    return mm._encode_image(imgs[0:1], None, normalize=False)[0]


def sched(mm, k=5):
    return regress_views(mm, feats[5], feats[:k], [True] * k, 0.0, H, Wd)


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a_enc = timeit(lambda: enc(m), iters)
a_s5 = timeit(lambda: sched(m, 5), iters)
a_s3 = timeit(lambda: sched(m, 3), iters)
a_s2 = timeit(lambda: sched(m, 2), iters)
print(f"A  encode {a_enc:.3f} ms | scheduler k=5 {a_s5:.3f} ms, k=3 {a_s3:.3f} ms, k=2 {a_s2:.3f} ms | serial enc+k5 {a_enc + a_s5:.3f}", flush=True)

s2 = torch.cuda.Stream()


def both():
    with torch.cuda.stream(s2):
        enc(m)
    sched(m, 5)


b = timeit(both, iters)
print(f"B  encode (stream 2) under scheduler k=5 (stream 1): {b:.3f} ms per pair of calls  (serial {a_enc + a_s5:.3f}; x{(a_enc + a_s5) / b:.2f})", flush=True)

m2 = make()
m3 = make()
torch.cuda.synchronize()


def worker(mm, fn, n, stream, out, idx):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn(mm)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn(mm)
        stream.synchronize()
        out[idx] = (time.perf_counter() - t0, n)


def run_threads(specs):
    out = [None] * len(specs)
    th = [threading.Thread(target=worker, args=(mm, fn, n, torch.cuda.Stream(), out, i)) for i, (mm, fn, n) in enumerate(specs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out, time.perf_counter() - t0


out, wall = run_threads([(m, lambda mm: sched(mm, 5), iters), (m2, lambda mm: sched(mm, 5), iters)])
rate = sum(n / dt for dt, n in out)
print(f"C  two scheduler chains (two handles, two threads): {rate:.1f} calls/s aggregate vs {1e3 / a_s5:.1f} alone  (x{rate * a_s5 / 1e3:.2f}); per-chain ms {[round(dt / n * 1e3, 3) for dt, n in out]}", flush=True)
out, wall = run_threads([(m, lambda mm: sched(mm, 5), iters), (m2, lambda mm: sched(mm, 5), iters), (m3, enc, 3 * iters)])
print(f"D  two scheduler chains + one encode chain: sched {sum(n / dt for dt, n in out[:2]):.1f} calls/s, encode {out[2][1] / out[2][0]:.1f} /s "
      f"(alone: {1e3 / a_s5:.1f}, {1e3 / a_enc:.1f}); per-chain ms {[round(dt / n * 1e3, 3) for dt, n in out]}", flush=True)
out, wall = run_threads([(m, enc, 3 * iters), (m2, enc, 3 * iters)])
print(f"E  two encode chains: {sum(n / dt for dt, n in out):.1f} /s aggregate vs {1e3 / a_enc:.1f} alone; per-chain ms {[round(dt / n * 1e3, 3) for dt, n in out]}", flush=True)
out, wall = run_threads([(m, enc, 3 * iters), (m2, enc, 3 * iters), (m3, enc, 3 * iters)])
print(f"F  three encode chains: {sum(n / dt for dt, n in out):.1f} /s aggregate; per-chain ms {[round(dt / n * 1e3, 3) for dt, n in out]}", flush=True)
