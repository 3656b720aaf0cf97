"""Where does a SLAM-scale GEMM spend its ~14 us?  In-kernel 100 MHz stamps of every workgroup (sta_bench_gemm_stamps)
against the HIP-event duration of the same launch.      python tools/gemm_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
st = torch.cuda.current_stream().cuda_stream
print(f"{'shape (M N K, r = in-place residual)':40s} {'WGs':>5s} {'ks':>3s} {'event':>7s} {'span':>7s} {'WG med':>7s} {'->tile0':>7s} {'loop':>7s} {'epilog':>7s} {'entries':>8s} {'exits':>7s}   (us)")
import numpy as np
CAP = 2048
for M, N, K, r in ((196, 3072, 1024, 0), (196, 1024, 1024, 1), (196, 4096, 1024, 0), (196, 1024, 4096, 1),
                   (1970, 2304, 768, 0), (1970, 768, 768, 1), (1970, 3072, 768, 0), (1970, 768, 3072, 1),
                   (12288, 1024, 4096, 2), (12288, 1024, 1024, 2), (12288, 4096, 1024, 0), (12288, 768, 768, 2)):
    for rep in range(2):
        out = (C.c_double * 10)()
        raw = (C.c_ulonglong * (4 * CAP))()
        _lib.check(m.lib.sta_bench_gemm_stamps(m._h, M, N, K, r, out, raw, CAP, st))
    o = list(out)
    print(f"{M:6d} {N:5d} {K:5d} {('r' if r == 1 else 'R' if r == 2 else ' '):24s} {int(o[0]):5d} {int(o[8]):3d} {o[7]:7.2f} {o[1]:7.2f} {o[9]:7.2f} {o[2]:7.2f} {o[3]:7.2f} {o[4]:7.2f} {o[5]:8.2f} {o[6]:7.2f}", flush=True)
    if M >= 12288:       # who finishes late?  lifetime of the workgroups by XCD (block b -> XCD b % 8) and by position in the grid
        a = np.array(raw, dtype=np.uint64).reshape(CAP, 4)[:int(o[0])].astype(np.int64)
        life = (a[:, 3] - a[:, 0]) * 0.01
        end = (a[:, 3] - a[:, 0].min()) * 0.01
        xcd = np.arange(len(life)) % 8
        print("      per XCD: median lifetime " + " ".join(f"{np.median(life[xcd == x]):6.1f}" for x in range(8)) +
              "  | last exit " + " ".join(f"{end[xcd == x].max():6.1f}" for x in range(8)))
        q = np.percentile(life, [0, 10, 50, 90, 100])
        print(f"      lifetime percentiles 0/10/50/90/100: {q[0]:.1f} {q[1]:.1f} {q[2]:.1f} {q[3]:.1f} {q[4]:.1f} us; loop {np.median((a[:,2]-a[:,1])*0.01):.1f}, epilogue p50/p90/max "
              f"{np.percentile((a[:,3]-a[:,2])*0.01, 50):.1f}/{np.percentile((a[:,3]-a[:,2])*0.01, 90):.1f}/{((a[:,3]-a[:,2])*0.01).max():.1f}")
