"""Same-process A/B of one experiment switch (sta_debug_set_option idx) on the benchmark workload (8 pairs @512x384; AB_B / AB_H / AB_W
in the environment pick another):
    python tools/ab_option.py idx v0 v1 [v2 ...] [--rounds R] [--prec P]
One model, the values alternate round-robin (box drift hits all arms equally); prints pairs/s per arm and the relative
difference of every arm's pts3d to arm 0's."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend

args = sys.argv[1:]
rounds, prec = 3, "f16x3h"
if "--rounds" in args: i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
if "--prec" in args: i = args.index("--prec"); prec = args[i + 1]; del args[i:i + 2]
idx, vals = int(args[0]), [int(v) for v in args[1:]]
B, H, Wd = int(os.environ.get("AB_B", 8)), int(os.environ.get("AB_H", 384)), int(os.environ.get("AB_W", 512))     # workload (default: the benchmark's)
m = STAFrontend(W.FULL, "cuda:0", precision=prec).load_procedural(seed=43)
imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
tot = {v: [] for v in vals}
outs = {}
for r in range(rounds):
    for v in vals:
        _lib.check(m.lib.sta_debug_set_option(m._h, idx, v))
        for _ in range(2):
            o = m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o = m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        outs[v] = o[0]["pts3d_pred"].float().clone()
        tot[v].append(B / dt)
        print(f"round {r} option[{idx}]={v}: {B / dt:7.2f} pairs/s  {dt * 1e3:7.2f} ms/step", flush=True)
base = sum(tot[vals[0]])
for v in vals:
    d = float((outs[v] - outs[vals[0]]).norm() / outs[vals[0]].norm())
    print(f"option[{idx}]={v}: mean {sum(tot[v]) / rounds:.2f} pairs/s  ratio to arm 0 {sum(tot[v]) / base:.4f}  rel diff of pts3d {d:.2e}")
