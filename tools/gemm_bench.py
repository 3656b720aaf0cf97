"""GEMM micro-benchmark on the GPU box: product shapes x tile families x ablations.

    python tools/gemm_bench.py [f16x3|f16]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vista_slam_amd import weights as W  # noqa: E402
from vista_slam_amd.sta_frontend import STAFrontend  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
m = STAFrontend(W.TINY, "cuda:0", precision=prec).load_procedural()
SHAPES = [("enc qkv", 12288, 3072, 1024), ("enc proj", 12288, 1024, 1024), ("enc fc1", 12288, 4096, 1024),
          ("enc fc2", 12288, 1024, 4096), ("dec fc1", 12304, 3072, 768), ("dec fc2", 12304, 768, 3072),
          ("dec proj", 12304, 768, 768), ("sq 8192", 8192, 8192, 8192)]
TILES = {1: "128x128", 2: "256x256", 3: "256x128"}
print(f"precision {prec}  (TFLOP/s algorithmic; x3 MFMA products in f16x3)")
for name, M, N, K in SHAPES:
    row = f"{name:9s} {M:6d}x{N:5d}x{K:5d} "
    for tile, tn in TILES.items():
        ms = m.bench_gemm(M, N, K, iters=10, tile=tile)
        row += f" {tn}: {2.0 * M * N * K / ms / 1e9:7.1f}"
    print(row, flush=True)
print("ablations on enc fc1 / 8192^3 (tile 256x256): 0 full, 1 no-DMA, 2 DMA-only, 3 MFMA-only")
for name, M, N, K in (SHAPES[2], SHAPES[-1]):
    row = f"{name:9s}"
    for abl in (0, 1, 2, 3):
        ms = m.bench_gemm(M, N, K, iters=10, tile=2, ablation=abl)
        row += f"  abl{abl}: {ms * 1e3:8.1f} us ({2.0 * M * N * K / ms / 1e9:7.1f} TF)"
    print(row, flush=True)
    row = f"{name:9s}"
    for abl in (0, 1, 2, 3):
        ms = m.bench_gemm(M, N, K, iters=10, tile=3, ablation=abl)
        row += f"  t3abl{abl}: {ms * 1e3:7.1f} us ({2.0 * M * N * K / ms / 1e9:7.1f} TF)"
    print(row, flush=True)
