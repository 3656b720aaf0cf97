"""Every committed reference golden in a given precision mode (GPU box): per-case worst relative error vs the 1e-3 bar."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as G
precs = [a for a in sys.argv[1:] if a != "--stress"] or ["f16x3h"]
STRESS = [f"tiny_48x80_sharp_s{sd}{sm}" for sd in (44, 45, 46, 47) for sm in ("_smooth", "")]
CASES = ["tiny_32x32_b1", "tiny_48x64_b2", "tiny_48x64_b2_sharp", "tiny_48x80_smooth_sharp", "full_224_b1", "full_224_b1_sharp", "full_384x512_b1"]
from vista_slam_amd import _lib
_orig_model = G.model


def _masked_model(mask):
    def f(cfg_name="tiny", qk_gain=1.0, precision="f16x3", seed=43):
        m = _orig_model(cfg_name, qk_gain, "f16x3", seed)
        _lib.check(m.lib.sta_set_mx_mask(m._h, mask))
        return m
    return f


if "--stress" in sys.argv:
    CASES = ["tiny_48x64_b2_sharp", "tiny_48x80_smooth_sharp"] + STRESS
for prec in precs:
    if prec.startswith("mask"):          # e.g. mask28 = fc1 + fc2 + head in the f16mx arithmetic, the rest f16x3
        G.model = _masked_model(int(prec[4:]))
    else:
        G.model = _orig_model
    for case in CASES:
        if case.startswith("full"):
            G.drop_models()
        r = G.run_golden_case(case, "f16x3" if prec.startswith("mask") else prec)
        k = max(r, key=r.get)
        print(f"{prec:8s} {case:26s} worst {r[k]:.2e} ({k})  {'ok' if r[k] < 1e-3 else 'FAIL'}", flush=True)
    G.drop_models()
