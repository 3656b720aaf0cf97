import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
M, N, K, tile = (int(x) for x in sys.argv[1:5])
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
ms = m.bench_gemm(M, N, K, iters=10, tile=tile)
print(f"{M}x{N}x{K} tile {tile}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF")
