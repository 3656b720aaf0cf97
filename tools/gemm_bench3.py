import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
m = STAFrontend(W.TINY, "cuda:0", precision=prec).load_procedural()
SHAPES = [("enc qkv", 12288, 3072, 1024), ("enc proj", 12288, 1024, 1024), ("enc fc1", 12288, 4096, 1024),
          ("enc fc2", 12288, 1024, 4096), ("dec fc1", 12304, 3072, 768), ("dec fc2", 12304, 768, 3072),
          ("dec proj", 12304, 768, 768), ("sq 8192", 8192, 8192, 8192)]
print("precision", prec, " TFLOP/s algorithmic per tile family")
for name, M, N, K in SHAPES:
    row = f"{name:9s} {M:6d}x{N:5d}x{K:5d} "
    for tile, tn in ((1, "128x128"), (2, "256x256/8w"), (3, "256x128/8w"), (4, "256x256/4w")):
        ms = m.bench_gemm(M, N, K, iters=10, tile=tile)
        row += f" {tn}: {2.0 * M * N * K / ms / 1e9:6.1f}"
    print(row, flush=True)
