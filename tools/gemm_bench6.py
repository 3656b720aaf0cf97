import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
SHAPES = [("enc qkv", 12288, 3072, 1024), ("enc proj", 12288, 1024, 1024), ("enc fc1", 12288, 4096, 1024),
          ("enc fc2", 12288, 1024, 4096), ("dec qkv", 12304, 2304, 768), ("dec fc1", 12304, 3072, 768), ("dec fc2", 12304, 768, 3072),
          ("dec proj", 12304, 768, 768)]
for name, M, N, K in SHAPES:
    row = f"{name:9s} {M:6d}x{N:5d}x{K:5d} "
    for tile, tn in ((0, "auto"), (1, "128x128"), (2, "256x256"), (3, "256x128"), (5, "192x256"), (6, "192x128")):
        ms = m.bench_gemm(M, N, K, iters=20, tile=tile)
        row += f" {tn}: {ms*1e3:6.1f}us"
    print(row, flush=True)
