"""Attention kernels alone (sta_bench_attention): the small-grid kernel (attention.h) vs the pipelined one (attention2.h) on the
encoder / decoder shapes of the benchmark.   python tools/attn_bench.py [precision]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd.sta_frontend import STAFrontend
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
m = STAFrontend(W.TINY, "cuda:0", precision=prec).load_procedural()
st = torch.cuda.current_stream().cuda_stream
for name, S, heads, nq, nk, pose in (("enc 16x16 768", 16, 16, 768, 768, 0), ("dec 16x12 768+pose", 16, 12, 768, 768, 1),
                                     ("enc 224^2 B8", 16, 16, 196, 196, 0), ("enc 2x16 768", 2, 16, 768, 768, 0)):
    row = f"{name:22s}"
    gf = 4.0 * S * heads * (nq + pose) * (nk + pose) * 64 / 1e9
    for rep in range(2):
        for which in (2, 1):
            ms = C.c_float()
            _lib.check(m.lib.sta_bench_attention(m._h, S, heads, nq, nk, pose, 20, which, C.byref(ms), st))
            row += f"  k{which}: {ms.value * 1e3:7.1f} us ({gf / ms.value:5.0f} TF)"
    print(row, flush=True)
if os.environ.get("STA_BENCH_EXPERIMENTS") == "1":      # component ablations of attention2.h's tile body (library built with the flag)
    names = {0: "full", 1: "no MFMA", 2: "no VALU items", 4: "no fragment reads", 8: "no DMA/barrier", 3: "reads+sync only", 14: "MFMA only", 7: "sync only", 15: "nothing"}
    for name, S, heads, nq, nk, pose in (("enc 16x16 768", 16, 16, 768, 768, 0),):
        row = f"{name:16s}"
        for abl, an in names.items():
            _lib.check(m.lib.sta_debug_set_option(m._h, 2, abl))
            ms = C.c_float()
            _lib.check(m.lib.sta_bench_attention(m._h, S, heads, nq, nk, pose, 20, 1, C.byref(ms), st))
            row += f"  {an}: {ms.value * 1e3:6.1f}"
        _lib.check(m.lib.sta_debug_set_option(m._h, 2, 0))
        print(row, flush=True)
