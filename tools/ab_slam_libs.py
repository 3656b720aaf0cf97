"""Same-process, same-box A/B of two builds of the library at SLAM scale (224x224, batch 1 encode; 5-edge scheduler call):
    python tools/ab_slam_libs.py [other.so] [rounds]
NEW = vista_slam_amd/libsta_mi355.so, OLD = tools/ab/libsta_old.so by default; the two frontends alternate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd.sta_frontend import STAFrontend
from vista_slam_amd.slam_scheduler import regress_views

other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "ab", "libsta_old.so")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
new = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
prod = _lib._lib
_lib._lib = _lib.load_other(other)
old = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
_lib._lib = prod
imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=7)).cuda()
res = {}
for r in range(rounds):
    for tag, m in (("NEW", new), ("OLD", old)):
        fa, _ = m._encode_image(imgs[:1], None, normalize=False)
        fb, _ = m._encode_image(imgs[1:], None, normalize=False)
        for name, fn in (("encode", lambda: m._encode_image(imgs[:1], None, normalize=False)),
                         ("sched5", lambda: regress_views(m, fa, [fb] * 5, [True] * 5, 0.0, 224, 224)),
                         ("sched2", lambda: regress_views(m, fa, [fb] * 2, [True] * 2, 0.0, 224, 224))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            res.setdefault((tag, name), []).append((time.perf_counter() - t0) / 20 * 1e3)
for name in ("encode", "sched5", "sched2"):
    a, b = min(res[("NEW", name)]), min(res[("OLD", name)])
    print(f"{name}: NEW {a:.3f} ms  OLD {b:.3f} ms  ratio (OLD/NEW) {b / a:.4f}")
