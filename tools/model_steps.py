"""A few whole-path steps (8 pairs @512x384; AB_B / AB_H / AB_W in the environment change that) with a forced GEMM tile family,
for rocprofv3 --kernel-trace.

    rocprofv3 --kernel-trace -d gpurun_out/prof_v9 -- python tools/model_steps.py 9 [steps] [precision]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd.sta_frontend import STAFrontend
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if variant != 0:
    _lib.use_test_hooks()      # a forced tile family needs the test-hooks build; variant 0 profiles the PRODUCT library
if os.environ.get("STA_AB_LIB"):      # profile ANOTHER build of the library (kernel-level A/B on one box: tools/r6/run_s.sh)
    _lib._lib = _lib.load_other(os.environ["STA_AB_LIB"])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
m = STAFrontend(W.FULL, "cuda:0", precision=prec).load_procedural(seed=43)
if variant != 0:
    _lib.check(m.lib.sta_set_gemm_variant(m._h, variant))
B, H, Wd = (int(os.environ.get(k, d)) for k, d in (("AB_B", 8), ("AB_H", 384), ("AB_W", 512)))      # workload: env AB_B / AB_H / AB_W
imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
for _ in range(steps):
    m.forward_pair(imgs[:B], imgs[B:])
torch.cuda.synchronize()
