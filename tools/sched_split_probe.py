"""Where a 5-edge scheduler call spends its time: phase A (gather + decode + pose heads) and phase B (accept / reject, DPT heads,
reductions) each timed alone behind a device synchronisation, against the whole call - the difference is what the host
synchronisation between the phases (slam.py:169) costs on top of the kernels.   python tools/sched_split_probe.py [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
from vista_slam_amd.slam_scheduler import regress_views, regress_views_begin, regress_views_finish
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=7)).cuda()
fa, _ = m._encode_image(imgs[:1], None, normalize=False)
fb, _ = m._encode_image(imgs[1:], None, normalize=False)
adj = [True] * k
for _ in range(5):
    regress_views(m, fa, [fb] * k, adj, 0.0, 224, 224)
torch.cuda.synchronize()
n = 40
ta = tb = tw = 0.0
for _ in range(n):
    t0 = time.perf_counter(); p = regress_views_begin(m, fa, [fb] * k, 224, 224); torch.cuda.synchronize(); t1 = time.perf_counter()
    regress_views_finish(m, p, adj, 0.0); torch.cuda.synchronize(); t2 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    regress_views(m, fa, [fb] * k, adj, 0.0, 224, 224)
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f"k={k}: phase A alone {ta / n * 1e3:.3f} ms  phase B alone {tb / n * 1e3:.3f} ms  sum {(ta + tb) / n * 1e3:.3f} ms   whole call (back to back) {tw / n * 1e3:.3f} ms")
