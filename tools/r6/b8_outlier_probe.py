"""Where does the B = 8 error of the outlier-statistics weights come from?  main_pts3d of slot 0 against the reference golden for: the
round-start build, the current build, the current build with the implicit-GEMM tail (switch 0), without the halo kernels (family 9)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import load_golden, rel_l2, max_rel
from vista_slam_amd import weights as W, _lib
from vista_slam_amd.sta_frontend import STAFrontend

case = sys.argv[1] if len(sys.argv) > 1 else "full_384x512_b1_outlier"
g, meta = load_golden(case)
H, Wd, sub, B = int(meta["H"]), int(meta["W"]), int(meta["sub"]), 8
im = W.synth_images(2, H, Wd, seed=int(meta["seed"]), tag=0)
extra = W.synth_images(2 * B, H, Wd, seed=int(meta["seed"]), tag=5)
a = torch.from_numpy(np.concatenate([im[:1], extra[:B - 1]])).cuda()
b = torch.from_numpy(np.concatenate([im[1:], extra[B:2 * B - 1]])).cuda()


def run(tag, lib, prec="f16x3h", opt=None, variant=None, nb=B):
    m = STAFrontend(W.FULL, "cuda:0", precision=prec, lib=lib).load_procedural(seed=int(meta["seed"]), qk_gain=float(meta["qk_gain"]), outlier=int(meta.get("outlier", 0)))
    if opt: _lib.check(m.lib.sta_debug_set_option(m._h, opt[0], opt[1]))
    if variant is not None: _lib.check(m.lib.sta_set_gemm_variant(m._h, variant))
    main, supp = m.forward_pair(a[:nb], b[:nb])
    torch.cuda.synchronize()
    e = {s: (rel_l2(o["pts3d_pred"][:1].cpu().numpy()[:, ::sub, ::sub], g[f"{s}_pts3d"]), rel_l2(o["conf"][:1].cpu().numpy()[:, ::sub, ::sub], g[f"{s}_conf"])) for s, o in (("main", main), ("supp", supp))}
    print(f"{tag:46s} B={nb} {prec}: main pts {e['main'][0]:.2e} conf {e['main'][1]:.2e} | supp pts {e['supp'][0]:.2e} conf {e['supp'][1]:.2e}  range {m.range_report()}", flush=True)
    del m; torch.cuda.empty_cache()


old = _lib.load_other(os.path.join(ROOT, "tools", "ab", "libsta_old.so"))
run("round-start build", old)
run("round-start build", old, nb=1)
run("current build", _lib.load())
run("current build", _lib.load(), nb=1)
run("current build", _lib.load(), prec="f16x3")
t = _lib.load_test()
run("current, implicit-GEMM tail (switch 0 = 1)", t, opt=(0, 1))
run("current, no halo kernels (family 9)", t, variant=9)
run("round-start test build, no halo kernels", _lib.load_other(os.path.join(ROOT, "tools", "ab", "libsta_old_test.so")), variant=9)
