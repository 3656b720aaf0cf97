"""Same-process, same-box A/B of two builds of the library on the benchmark workload (8 pairs @512x384):
    python tools/ab_inproc.py [other.so] [precision] [rounds]
NEW = vista_slam_amd/libsta_mi355.so, OLD = tools/ab/libsta_old.so by default (kept out of git); AB_B / AB_H / AB_W in the
environment change the workload (e.g. AB_B=2, or AB_H=224 AB_W=224).  The two frontends
live side by side (own weights, own workspace) and alternate, so box drift hits both arms equally."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd.sta_frontend import STAFrontend

other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "ab", "libsta_old.so")
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3h"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B, H, Wd = (int(os.environ.get(k, d)) for k, d in (("AB_B", 8), ("AB_H", 384), ("AB_W", 512)))      # workload: env AB_B / AB_H / AB_W
new = STAFrontend(W.FULL, "cuda:0", precision=prec).load_procedural(seed=43)
prod = _lib._lib
_lib._lib = _lib.load_other(other)
old = STAFrontend(W.FULL, "cuda:0", precision=prec).load_procedural(seed=43)
_lib._lib = prod
imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
outs = {}
tot = {"NEW": [], "OLD": []}
for r in range(rounds):
    for tag, m in (("NEW", new), ("OLD", old)):
        for _ in range(2):
            o = m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o = m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        outs[tag] = o
        tot[tag].append(B / dt)
        print(f"round {r} {tag}: {B / dt:7.2f} pairs/s  {dt * 1e3:7.2f} ms/step", flush=True)
a, b = outs["NEW"][0]["pts3d_pred"].float(), outs["OLD"][0]["pts3d_pred"].float()
print(f"mean NEW {sum(tot['NEW']) / rounds:.2f}  OLD {sum(tot['OLD']) / rounds:.2f}  ratio {sum(tot['NEW']) / sum(tot['OLD']):.4f}  "
      f"rel diff of the two builds' pts3d {float((a - b).norm() / b.norm()):.2e}")
