import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
M, N, K = 12288, 1024, 4096
for iters in (5, 20, 100, 400, 400):
    ms = m.bench_gemm(M, N, K, iters=iters, tile=2)
    print(f"fc2 tile2 iters={iters:4d}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TF", flush=True)
M, N, K = 12288, 4096, 1024
for iters in (5, 400):
    ms = m.bench_gemm(M, N, K, iters=iters, tile=2)
    print(f"fc1 tile2 iters={iters:4d}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TF", flush=True)
