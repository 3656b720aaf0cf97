import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
for (M, N) in ((12288, 4096), (12288, 1024), (12288, 3072)):
    for tile in (2, 3, 4):
        row = f"M={M} N={N} tile={tile}: "
        for K in (256, 512, 1024, 2048, 4096):
            ms = m.bench_gemm(M, N, K, iters=30, tile=tile)
            row += f" K={K}:{ms*1e3:7.1f}us"
        print(row, flush=True)
