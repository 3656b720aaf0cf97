"""Whole-path A/B on one box: (precision, forced GEMM family) configurations, two interleaved passes.
    python tools/model_ab.py f16x3:0 f16x3:4 f16x3h:0     (optional third field: an STA_EXPERIMENT_<name> environment switch to set)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
cfgs = [(c.split(":")[0], int(c.split(":")[1]), (c.split(":") + [""])[2]) for c in (sys.argv[1:] or ["f16x3h:0", "f16x3:0"])]   # precision:forced tile family[:ENVVAR to set]
m = STAFrontend(W.FULL, "cuda:0", precision="f16x3").load_procedural(seed=43)
B, H, Wd = 8, 384, 512
imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
for rep in range(2):
    for prec, v, envv in cfgs:
        for k in [k for k in os.environ if k.startswith("STA_EXPERIMENT_")]:
            del os.environ[k]
        if envv:
            os.environ["STA_EXPERIMENT_" + envv] = "1"
        m.set_precision(prec)
        _lib.check(m.lib.sta_set_gemm_variant(m._h, v))
        for _ in range(2):
            m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        print(f"pass {rep} {prec}:{v}:{envv}: {B / dt:7.2f} pairs/s  {dt * 1e3:7.2f} ms/step", flush=True)
