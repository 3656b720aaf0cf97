"""Ablation matrix of the gemm2 main loop (bench-only kernel variants; bit mask: 1 no DMA, 2 no LDS
fragment reads, 4 no MFMA).  tile 2 = 256x256 (1 workgroup/CU), tile 6 = 192x128 (2 workgroups/CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
NAMES = {0: "full", 1: "noDMA", 2: "noLDS", 3: "MFMA only", 4: "noMFMA", 5: "LDS only", 6: "DMA only", 7: "barriers only"}
for name, M, N, K in (("enc fc1", 12288, 4096, 1024), ("sq 8192", 8192, 8192, 8192)):
    for tile in (2, 6):
        m.bench_gemm(M, N, K, iters=3, tile=tile)
        row = f"{name} tile{tile}:"
        for abl in range(8):
            ms = m.bench_gemm(M, N, K, iters=10, tile=tile, ablation=abl)
            row += f" {NAMES[abl]}={ms*1e3:.0f}us"
        print(row, flush=True)
