"""Same-process A/B of one experiment switch (sta_debug_set_option idx) at SLAM scale (224x224, B = 1 encode; 5-edge scheduler):
    python tools/ab_slam.py idx v0 v1 [v2 ...] [--rounds R]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
from vista_slam_amd.slam_scheduler import regress_views

args = sys.argv[1:]
rounds = 3
if "--rounds" in args: i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
idx, vals = int(args[0]), [int(v) for v in args[1:]]
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=7)).cuda()
fa, pa = m._encode_image(imgs[:1], None, normalize=False)
fb, pb = m._encode_image(imgs[1:], None, normalize=False)
stages = {"encode": lambda: m._encode_image(imgs[:1], None, normalize=False),
          "sched5": lambda: regress_views(m, fa, [fb] * 5, [True] * 5, 0.0, 224, 224)}
tot = {(v, k): [] for v in vals for k in stages}
ref = {}
for r in range(rounds):
    for v in vals:
        _lib.check(m.lib.sta_debug_set_option(m._h, idx, v))
        for k, fn in stages.items():
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                out = fn()
            torch.cuda.synchronize()
            tot[(v, k)].append((time.perf_counter() - t0) / 20 * 1e3)
            if k == "encode":
                f = out[0].float().clone()
                ref.setdefault("f", f)
                d = float((f - ref["f"]).norm() / ref["f"].norm())
        print(f"round {r} option[{idx}]={v}: encode {tot[(v, 'encode')][-1]:.3f} ms  sched5 {tot[(v, 'sched5')][-1]:.3f} ms  rel diff of features vs arm 0 {d:.1e}", flush=True)
for v in vals:
    print(f"option[{idx}]={v}: encode {min(tot[(v, 'encode')]):.3f} ms (min)  sched5 {min(tot[(v, 'sched5')]):.3f} ms (min)")
