"""Ablations of the small-grid (ring) GEMM family inside the product flow: libraries built with -DSTA_RING_ABL=n
(1 = no DMA, 2 = no LDS fragment reads, 4 = no MFMA; sums combine) report the in-kernel stamps of the same launches.
    for a in 0 1 2 4 6 7; do hipcc ... -DSTA_DEV_FAST -DSTA_RING_ABL=$a -o vista_slam_amd/libsta_abl$a.so csrc/sta_api.hip; done
    python tools/ring_ablate.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
NAMES = {0: "full", 1: "no DMA", 2: "no LDS reads", 4: "no MFMA", 6: "DMA only", 7: "barriers only", 3: "MFMA only", 5: "LDS reads only"}
SHAPES = ((196, 3072, 1024, 0), (196, 1024, 1024, 1), (196, 1024, 4096, 1), (392, 1024, 4096, 1), (1970, 768, 3072, 1))
st = torch.cuda.current_stream().cuda_stream
prod = _lib._lib
print(f"{'variant':16s} " + " ".join(f"{f'{M}x{N}x{K}' + ('r' if r else ''):>26s}" for M, N, K, r in SHAPES) + "    (K tiles per workgroup | loop us | us per K tile)")
for a in (0, 1, 2, 4, 6, 7):
    path = os.path.join(ROOT, "vista_slam_amd", f"libsta_abl{a}.so")
    if not os.path.exists(path):
        continue
    _lib._lib = _lib.load_other(path)
    m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
    row = f"{NAMES[a]:16s} "
    for M, N, K, r in SHAPES:
        for rep in range(3):
            out = (C.c_double * 10)()
            raw = (C.c_ulonglong * (4 * 2048))()
            _lib.check(m.lib.sta_bench_gemm_stamps(m._h, M, N, K, r, out, raw, 2048, st))
        o = list(out)
        ks = max(1, int(o[8])); nkt = K // 32 // ks
        row += f"{nkt:4d} | {o[3]:6.2f} | {o[3] / nkt:5.3f}      "
    print(row, flush=True)
    _lib._lib = prod
