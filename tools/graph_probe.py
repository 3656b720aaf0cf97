"""Does a HIP graph shorten the SLAM-scale chains on the GPU side?  Capture sta_encode (224x224, batch 1: ~220 dependent dispatches)
and the scheduler's first phase with torch.cuda.CUDAGraph and compare replay with eager launches.   python tools/graph_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend

m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=7)).cuda()
img = imgs[:1].contiguous()
for _ in range(3):
    ref = m._encode_image(img, None, normalize=False)[0]
torch.cuda.synchronize()


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timeit(lambda: m._encode_image(img, None, normalize=False))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        m._encode_image(img, None, normalize=False)          # this stream's scratch context exists before the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = m._encode_image(img, None, normalize=False)[0]
graph = timeit(g.replay)
torch.cuda.synchronize()
print(f"encode: eager {eager:.3f} ms   graph replay {graph:.3f} ms   identical output: {bool(torch.equal(out, ref))}")
