import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
for name, M, N, K in (("enc fc1", 12288, 4096, 1024), ("enc fc2", 12288, 1024, 4096), ("sq 8192", 8192, 8192, 8192)):
    row = f"{name:9s}"
    for abl in (0, 4, 1, 2, 6, 3, 7):
        ms = m.bench_gemm(M, N, K, iters=10, tile=2, ablation=abl)
        row += f" abl{abl}:{ms*1e3:8.1f}us({2.0*M*N*K/ms/1e9:6.1f})"
    print(row, flush=True)
