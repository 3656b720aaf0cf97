"""SLAM-scale (224x224, batch 1) calls for rocprofv3: wall time per call vs the sum of kernel durations
tells how launch-bound the split entry points are."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
from vista_slam_amd.slam_scheduler import regress_views
m = STAFrontend(W.FULL, "cuda:0", precision="f16x3").load_procedural(seed=43)
imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=7)).cuda()
ts = torch.tensor([[224, 224]])
which = sys.argv[1] if len(sys.argv) > 1 else "encode"
iters = 20
fa, pa = m._encode_image(imgs[:1], ts, normalize=False)
fb, pb = m._encode_image(imgs[1:], ts, normalize=False)
fn = {"encode": lambda: m._encode_image(imgs[:1], ts, normalize=False),
      "decode": lambda: m._decode_stereo(fa, fb, pa, pb),
      "sched5": lambda: regress_views(m, fa, [fb] * 5, [True] * 5, 0.0, 224, 224)}[which]
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    fn()
torch.cuda.synchronize()
print(f"{which}: wall {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/call over {iters} calls (+3 warm-up calls, +setup encodes)")
