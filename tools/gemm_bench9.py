import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
for name, M, N, K in (("warm", 12288, 4096, 1024), ("enc fc1", 12288, 4096, 1024), ("enc fc2", 12288, 1024, 4096), ("enc qkv", 12288, 3072, 1024), ("sq8192", 8192, 8192, 8192)):
    row = f"{name:9s}"
    for tile, tn in ((2, "256x256"), (11, "256x256 stagger"), (6, "192x128"), (12, "192x128 stagger")):
        ms = m.bench_gemm(M, N, K, iters=20, tile=tile)
        row += f"  {tn}: {ms*1e3:7.1f}us ({2.0*M*N*K/ms/1e9:5.1f})"
    print(row, flush=True)
