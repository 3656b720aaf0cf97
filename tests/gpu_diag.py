"""Diagnostic runner for the GPU box: runs every parity check in both precisions without stopping
at the first failure and writes a report to gpurun_out/diag.log.

    python tests/gpu_diag.py [kernels] [tiny] [full]
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "diag.log"), "a")


def out(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


def run(name, fn, *args, **kw):
    t = time.time()
    try:
        r = fn(*args, **kw)
        out(f"[{name}] " + " ".join(f"{k}={v:.3e}" for k, v in r.items()) + f"  ({time.time() - t:.1f}s)")
    except Exception as e:   # noqa: BLE001
        out(f"[{name}] EXCEPTION {type(e).__name__}: {e}")
        traceback.print_exc()


def main():
    import torch
    import gpu_checks as G
    sel = sys.argv[1:] or ["kernels", "tiny", "full"]
    out("==== diag", time.ctime(), torch.cuda.get_device_name(0), sel)
    for prec in ("f16x3", "f16", "f16x3h"):
        if "kernels" in sel:
            run(f"gemm f32-epi {prec}", G.check_gemm, prec)
            run(f"gemm f32-epi resid {prec}", G.check_gemm, prec, resid=True)
            run(f"gemm big {prec}", G.check_gemm, prec, M=520, N=384, K=1024)
            run(f"gemm f16-epi gelu {prec}", G.check_gemm, prec, act=1, via_f16=1)
            run(f"gemm f16-epi relu {prec}", G.check_gemm, prec, act=2, via_f16=1)
            run(f"qkv_rope pose {prec}", G.check_qkv_rope, prec)
            run(f"qkv_rope enc {prec}", G.check_qkv_rope, prec, hp=14, wp=14, pose_tok=0, S=1)
            run(f"attn self 197 {prec}", G.check_attention, prec)
            run(f"attn cross 197 {prec}", G.check_attention, prec, kv_shift=1)
            run(f"attn 70x130 sharp {prec}", G.check_attention, prec, nq=70, nk=130, sharp=6.0)
            run(f"attn 769 {prec}", G.check_attention, prec, S=1, heads=1, nq=769, nk=769, sharp=3.0)
            run(f"conv3 s1 {prec}", G.check_conv3, prec)
            run(f"conv3 s2 odd {prec}", G.check_conv3, prec, stride=2)
            run(f"conv3 s2 even {prec}", G.check_conv3, prec, stride=2, H=6, W_=8)
            run(f"conv3 relu_in+relu+resid {prec}", G.check_conv3, prec, relu_in=1, act=2, resid=True, Cin=96, Co=256, H=9, W_=12)
            run(f"convt k4 {prec}", G.check_convt, prec)
            run(f"convt k2 {prec}", G.check_convt, prec, Cdim=192, k=2)
            run(f"up2 {prec}", G.check_up2, prec)
            run(f"up2 crop {prec}", G.check_up2, prec, H=2, W_=3, crop=(3, 5))
            run(f"layernorm 768 {prec}", G.check_layernorm, prec)
            run(f"layernorm 128 {prec}", G.check_layernorm, prec, Cdim=128)
            run(f"layernorm 1024 {prec}", G.check_layernorm, prec, Cdim=1024)
            run(f"ops golden {prec}", G.check_ops_golden, prec)
        if "tiny" in sel:
            for case in ("tiny_32x32_b1", "tiny_48x64_b2", "tiny_48x64_b2_sharp", "tiny_48x80_smooth_sharp"):
                run(f"{case} {prec}", G.run_golden_case, case, prec)
    if "full" in sel:
        G.drop_models()
        for case in ("full_224_b1", "full_384x512_b1"):
            for prec in ("f16x3", "f16", "f16x3h"):
                run(f"{case} {prec}", G.run_golden_case, case, prec)
        G.drop_models()
        for prec in ("f16x3", "f16", "f16x3h"):
            run(f"full_224_b1_sharp {prec}", G.run_golden_case, "full_224_b1_sharp", prec)
    out("==== done")


if __name__ == "__main__":
    main()
