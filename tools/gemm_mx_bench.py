"""Shipped f16mx GEMM kernel (tile 16; 17 = with the in-place residual) vs the f16x3 kernel (tile 6 / 13) and the
instruction-mix what-if (tile 15), 192x128 tiles, model shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
for name, M, N, K in (("enc fc1", 12288, 4096, 1024), ("enc fc2", 12288, 1024, 4096), ("enc proj", 12288, 1024, 1024), ("enc qkv-shape", 12288, 3072, 1024),
                      ("dec fc1", 12304, 3072, 768), ("dec proj", 12304, 768, 768)):
    row = f"{name:14s}"
    for tile in (6, 15, 16, 13, 17):
        ms = m.bench_gemm(M, N, K, iters=20, tile=tile)
        ghz = m.lib.sta_bench_gemm_last_ghz()
        row += f"  t{tile}: {ms*1e3:6.1f} us @{ghz:.2f}"
    print(row, flush=True)
