"""Which streams the pipelined slam_replay runs on matters (hardware-queue placement): the same triple of streams is slow or fast
every time it is used.  python tools/queue_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
dev = torch.device("cuda:0")
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
pool = [torch.cuda.Stream(device=dev) for _ in range(12)]
for name, idx in (("0,1,2", (0, 1, 2)), ("0,1,2", (0, 1, 2)), ("3,4,5", (3, 4, 5)), ("0,1,2", (0, 1, 2)), ("1,2,3", (1, 2, 3)), ("4,5,6", (4, 5, 6)), ("0,2,4", (0, 2, 4)), ("1,5,9", (1, 5, 9)), ("0,4,8", (0, 4, 8))):
    r = bench.slam_replay(m, dev, frames=60, streams=[pool[i] for i in idx])
    print(f"streams {name}: pipelined {r['keyframes_per_s']:.1f} (first pass {r['first_pass_keyframes_per_s']:.1f})  single {r['single_stream']['keyframes_per_s']:.1f}", flush=True)
