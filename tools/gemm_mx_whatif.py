"""What would an 'f16 main product + one MX-fp8 correction MFMA' scheme buy?  Bench-only kernels with that instruction
mix (tiles 14 = 256x256, 15 = 192x128; the outputs are not a GEMM) against the shipped f16x3 kernels (tiles 2 / 6)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
for name, M, N, K in (("sq 8192", 8192, 8192, 8192), ("enc fc1", 12288, 4096, 1024), ("enc fc2", 12288, 1024, 4096), ("enc qkv-shape", 12288, 3072, 1024)):
    for base, mx in ((2, 14), (6, 15)):
        row = f"{name:14s}"
        for tile in (base, mx, base, mx):
            ms = m.bench_gemm(M, N, K, iters=20, tile=tile)
            ghz = m.lib.sta_bench_gemm_last_ghz()
            row += f"  tile{tile}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF-eq @{ghz:.2f} GHz"
        print(row, flush=True)
