"""In-model duration of the dominant GEMM family per shape (HIP events around every launch inside the full forward)
vs the same shapes in the back-to-back micro-benchmark (same box, same tile, in-place residual)."""
import os, sys, collections, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.FULL, "cuda:0", precision="f16x3").load_procedural(seed=43)
B, H, Wd = 8, 384, 512
imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
for _ in range(3):
    m.forward_pair(imgs[:B], imgs[B:])
torch.cuda.synchronize()
m.kernel_timing(True)
for _ in range(4):
    m.forward_pair(imgs[:B], imgs[B:])
torch.cuda.synchronize()
cap = 4096
fl = (C.c_double * cap)(); ms = (C.c_float * cap)(); var = (C.c_int * cap)(); n = C.c_int()
_lib.check(m.lib.sta_kernel_timing_dump(m._h, cap, fl, ms, var, C.byref(n)))
m.kernel_timing(False)
groups = collections.defaultdict(list)
for i in range(n.value):
    groups[(round(fl[i] / 1e9, 2), var[i])].append(ms[i] * 1e3)
SHAPES = {25.77: (12288, 1024, 1024), 103.08: (12288, 1024, 4096), 19.33: (12288, 1024, 768), 14.51: (12304, 768, 768),
          58.06: (12304, 768, 3072), 19.36: (12304, 768, 1024)}
print(f"{'GF':>8s} {'var':>3s} {'n':>4s} {'in-model us':>12s} {'TF':>7s} | {'micro us':>9s} {'TF':>7s}  shape")
for (gf, v), ts in sorted(groups.items()):
    avg = sum(ts) / len(ts)
    shape = min(SHAPES.items(), key=lambda kv: abs(kv[0] - gf))
    line = f"{gf:8.2f} {v:3d} {len(ts):4d} {avg:12.1f} {gf / avg * 1e3:7.1f} |"
    if abs(shape[0] - gf) < 0.2 and v == 5:
        M, N, K = shape[1]
        mu = m.bench_gemm(M, N, K, iters=20, tile=13) * 1e3
        line += f" {mu:9.1f} {gf / mu * 1e3:7.1f}  {M}x{N}x{K}"
    print(line, flush=True)
