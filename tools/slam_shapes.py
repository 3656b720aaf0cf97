"""Per-launch GEMM / conv durations at the SLAM scale (224x224, batch 1; split entry points as slam.py calls them) and for
the batched keyframe scheduler (5 edges): where the time of the latency-bound regime goes."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
from vista_slam_amd.slam_scheduler import regress_views
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=7)).cuda()
ts = torch.tensor([[224, 224]])
epi = {0: "f32", 1: "f16", 2: "qkv", 3: "convT", 4: "gelu", 5: "f32r"}


def stage(name, fn, reps=3):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    m.kernel_timing(2)
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    recs = m.kernel_timing_records()
    m.kernel_timing(False)
    g = collections.OrderedDict()
    for (M, N, K, e, a, mx, fam, ms) in recs:
        g.setdefault((M, N, K, e, a, mx, fam), []).append(ms * 1e3)
    tot = sum(sum(v) for v in g.values()) / reps
    print(f"== {name}: wall {wall:.3f} ms, GEMM/conv launches {len(recs) // reps}, sum of their durations {tot / 1e3:.3f} ms")
    for (M, N, K, e, a, mx, fam), v in g.items():
        n = len(v) // reps
        print(f"   {M:7d} x{N:5d} x{K:5d} {epi[e]:>5s} {'conv' if a else 'dns':>4s} mx{mx} fam{fam}: {n:3d} x {sum(v) / len(v):7.1f} us = {n * sum(v) / len(v) / 1e3:6.3f} ms")
    return out


fa, pa = stage("encode B=1", lambda: m._encode_image(imgs[:1], ts, normalize=False))
fb, pb = m._encode_image(imgs[1:], ts, normalize=False)
d1, d2 = stage("decode B=1", lambda: m._decode_stereo(fa, fb, pa, pb))
stage("head_pts B=1", lambda: m.head_pts([fa] + [t[:, 1:, :] for t in d1], ts))
stage("scheduler 5 edges", lambda: regress_views(m, fa, [fb] * 5, [True] * 5, 0.0, 224, 224))
