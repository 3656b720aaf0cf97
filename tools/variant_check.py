"""Correctness of forced GEMM tile families on the GPU box: kernel-level GEMM checks (several K: odd / even tile counts,
ragged M) and reference goldens, for each variant id given on the command line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
import gpu_checks as G
variants = [int(x) for x in sys.argv[1:]] or [10, 11, 12]
for v in variants:
    worst = 0.0
    for (M, N, K) in ((300, 256, 32), (300, 256, 64), (300, 256, 96), (1000, 512, 128), (777, 256, 160), (192, 1024, 1024)):
        for kw in (dict(), dict(via_f16=1, act=1), dict(resid=True)):
            r = G.check_gemm("f16x3", M=M, N=N, K=K, variant=v, **kw)
            worst = max(worst, r["rel_l2"])
            if r["rel_l2"] > 1e-5:
                print(f"  variant {v} GEMM {M}x{N}x{K} {kw}: {r}")
    print(f"variant {v}: worst GEMM rel_l2 {worst:.2e}", flush=True)
    for case in ("tiny_48x64_b2", "tiny_48x80_smooth_sharp"):
        r = G.run_golden_case(case, "f16x3", variant=v)
        print(f"variant {v} {case}: max err {max(r.values()):.2e}", flush=True)
G.drop_models()
r = G.run_golden_case("full_384x512_b1", "f16x3", variant=variants[0])
print(f"variant {variants[0]} full_384x512_b1: max err {max(r.values()):.2e}")
