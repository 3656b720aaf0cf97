"""Same-process A/B of one experiment switch (sta_debug_set_option idx) on bench.slam_replay (224x224 keyframes, pipelined and
single-stream schedules):
    python tools/ab_replay.py idx v0 v1 [...] [--rounds R] [--frames F]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend

args = sys.argv[1:]
rounds, frames = 3, 120
if "--rounds" in args: i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
if "--frames" in args: i = args.index("--frames"); frames = int(args[i + 1]); del args[i:i + 2]
idx, vals = int(args[0]), [int(v) for v in args[1:]]
dev = torch.device("cuda:0")
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
acc = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        _lib.check(m.lib.sta_debug_set_option(m._h, idx, v))
        res = bench.slam_replay(m, dev, frames=frames)
        acc[v].append((res["keyframes_per_s"], res["single_stream"]["keyframes_per_s"]))
        print(f"round {r} option[{idx}]={v}: pipelined {acc[v][-1][0]:.1f} kf/s  single stream {acc[v][-1][1]:.1f} kf/s  same result {res['same_result_as_single_stream']}", flush=True)
for v in vals:
    print(f"option[{idx}]={v}: pipelined max {max(a for a, _ in acc[v]):.1f} mean {sum(a for a, _ in acc[v]) / rounds:.1f} kf/s   single max {max(b for _, b in acc[v]):.1f} mean {sum(b for _, b in acc[v]) / rounds:.1f} kf/s")
