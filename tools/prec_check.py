"""Every committed reference golden in a given precision mode (GPU box): per-case worst relative error vs the 1e-3 bar.
    python tools/prec_check.py [--stress | --outlier | --fullstress] prec ...
    STA_CHECKPOINT=FILE python tools/prec_check.py --checkpoint prec ...     # the real-checkpoint fixtures tests/golden/ckpt_*.npz
                                                                              # (oracle/gen_golden.py --checkpoint FILE) on the file's weights"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as G
precs = [a for a in sys.argv[1:] if not a.startswith("--")] or ["f16x3h"]
STRESS = [f"tiny_48x80_sharp_s{sd}{sm}" for sd in (44, 45, 46, 47) for sm in ("_smooth", "")]
CASES = ["tiny_32x32_b1", "tiny_48x64_b2", "tiny_48x64_b2_sharp", "tiny_48x80_smooth_sharp", "full_224_b1", "full_224_b1_sharp", "full_384x512_b1"]
OUTLIER = ["tiny_48x64_b2_outlier", "tiny_48x80_outlier_sharp", "full_224_b1_outlier", "tiny_48x64_b1_overflow"]
if "--stress" in sys.argv:
    CASES = ["tiny_48x64_b2_sharp", "tiny_48x80_smooth_sharp"] + STRESS
if "--outlier" in sys.argv:
    CASES = OUTLIER
if "--fullstress" in sys.argv:      # round 4: peaky attention / checkpoint-like ranges on the FULL architecture, incl. the headline resolution
    CASES = ["full_384x512_b1_sharp", "full_384x512_b1_outlier", "full_224_b1_sharp", "full_224_b1_sharp_s44_smooth", "full_224_b1_sharp_s45", "full_224_b1_outlier"]
precs = [p_ for p_ in precs if p_ != "--outlier"]
ckpt_fe = None
if "--checkpoint" in sys.argv:
    import glob, torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.sta_frontend import STAFrontend
    path = os.environ.get("STA_CHECKPOINT", "")
    assert path and os.path.exists(path), "set STA_CHECKPOINT to the checkpoint file"
    CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "ckpt_*.npz")))
    assert CASES, "no tests/golden/ckpt_*.npz: run `python oracle/gen_golden.py --checkpoint $STA_CHECKPOINT` in the build container first"
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    ckpt_fe = STAFrontend(W.FULL, "cuda:0").load_state_dict(sd, strict=True)
    fp = W.state_dict_fingerprint(sd)
    for c in CASES:
        g, _m = G.load_golden(c)
        assert str(g["ckpt_fingerprint"]) == fp, f"{c} was generated from another checkpoint than {path}"
for prec in precs:
    for case in CASES:
        if case.startswith("full"):
            G.drop_models()
        r = G.run_golden_case(case, prec, frontend=ckpt_fe)
        l2 = {k_: v for k_, v in r.items() if not k_.endswith("_maxrel")}
        mx = {k_: v for k_, v in r.items() if k_.endswith("_maxrel")}
        k = max(l2, key=l2.get); km = max(mx, key=mx.get)
        worst = max(r[k], r[km])
        print(f"{prec:8s} {case:28s} worst rel-L2 {r[k]:.2e} ({k})  worst max-abs/max-abs {r[km]:.2e} ({km})  {'ok' if worst < 1e-3 else 'FAIL'}  "
              f"range report (fp16, fp8 events): {G.last_range}", flush=True)
    G.drop_models()
