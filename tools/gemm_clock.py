"""Effective shader clock inside the GEMM main loop (DVFS: the chip clocks to its power budget).
In-kernel probe: s_memtime cycles per constant-100-MHz s_memrealtime tick (sta_bench_gemm_last_ghz)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
NAMES = {0: "full", 1: "noDMA", 2: "noLDS", 3: "MFMA only", 4: "noMFMA", 5: "LDS only", 6: "DMA only", 7: "barriers only"}
for name, M, N, K in (("sq 8192", 8192, 8192, 8192), ("enc fc1", 12288, 4096, 1024), ("enc fc2", 12288, 1024, 4096)):
    for tile in (2, 6):
        m.bench_gemm(M, N, K, iters=3, tile=tile)
        for abl in (0, 3, 1, 6):
            ms = m.bench_gemm(M, N, K, iters=20, tile=tile, ablation=abl)
            ghz = m.lib.sta_bench_gemm_last_ghz()
            tf = 2.0 * M * N * K / ms / 1e9
            print(f"{name} tile{tile} {NAMES[abl]:10s} {ms*1e3:8.0f} us  {tf:7.1f} TF alg  clock {ghz:.3f} GHz"
                  + (f"  issued/peak@clock {3*tf/(2500*ghz/2.4):.3f}" if abl in (0, 3, 1) else ""), flush=True)
