"""Host-side probe (GPU box): torch-CPU port timing vs thread count, each configuration in its own process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from oracle import torch_cpu as T
from vista_slam_amd import weights as Wt
n = int(sys.argv[1]); H, W_ = int(sys.argv[2]), int(sys.argv[3])
torch.set_num_threads(n)
sd = Wt.state_dict(Wt.FULL, seed=43)
imgs = Wt.synth_images(2, H, W_, seed=43, tag=0)
print(f"threads {n} {W_}x{H} cpu_count {os.cpu_count()}", flush=True)
for i in range(3):
    t0 = time.perf_counter(); T.forward_pair(Wt.FULL, sd, imgs[:1], imgs[1:]); print(f"  pass {i}: {time.perf_counter() - t0:.2f} s", flush=True)
