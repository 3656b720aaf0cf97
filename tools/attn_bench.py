"""The attention kernel alone (sta_bench_attention) on the encoder / decoder shapes of the benchmark.
    python tools/attn_bench.py [precision]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
m = STAFrontend(W.TINY, "cuda:0", precision=prec).load_procedural()
st = torch.cuda.current_stream().cuda_stream
for name, S, heads, nq, nk, pose in (("enc 16x16 768", 16, 16, 768, 768, 0), ("dec 16x12 768+pose", 16, 12, 768, 768, 1),
                                     ("dec 16x12 769 (round 2)", 16, 12, 769, 769, 0), ("enc 224^2 B8", 16, 16, 196, 196, 0), ("enc 2x16 768", 2, 16, 768, 768, 0),
                                     ("SLAM enc 1x16 196", 1, 16, 196, 196, 0), ("SLAM dec 10x12 196+pose", 10, 12, 196, 196, 1),
                                     ("SLAM dec 10x12 197 (round 2)", 10, 12, 197, 197, 0), ("SLAM dec 2x12 196+pose", 2, 12, 196, 196, 1),
                                     ("B1 dec 2x12 768+pose", 2, 12, 768, 768, 1), ("B1 dec 2x12 769 (round 2)", 2, 12, 769, 769, 0),
                                     ("B2 dec 4x12 768+pose", 4, 12, 768, 768, 1), ("B2 dec 4x12 769 (round 2)", 4, 12, 769, 769, 0)):
    row = f"{name:24s}"
    gf = 4.0 * S * heads * (nq + pose) * (nk + pose) * 64 / 1e9
    for rep in range(2):
        ms = C.c_float()
        _lib.check(m.lib.sta_bench_attention(m._h, S, heads, nq, nk, pose, 20, 0, C.byref(ms), st))
        row += f"  {ms.value * 1e3:7.1f} us ({gf / ms.value:5.0f} TF)"
    print(row, flush=True)
