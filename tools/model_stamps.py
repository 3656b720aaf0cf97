"""In-kernel stamps of every GEMM / convolution launch of one whole forward (sta_kernel_timing(h, 4)): per kernel class and shape the
HIP-event duration next to the in-kernel span and the median workgroup's phases (entry -> first K tile, main loop, epilogue).
    python tools/model_stamps.py            # 8 pairs @512x384;  AB_B / AB_H / AB_W change the workload"""
import collections, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
B, H, Wd = (int(os.environ.get(k, d)) for k, d in (("AB_B", 8), ("AB_H", 384), ("AB_W", 512)))
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
if os.environ.get("STA_TOOL_OPT"):        # "idx:value" -> sta_debug_set_option (e.g. 3:1 = LayerNorm fold off)
    i_, v_ = os.environ["STA_TOOL_OPT"].split(":"); _lib.check(m.lib.sta_debug_set_option(m._h, int(i_), int(v_)))
imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
for _ in range(2):
    m.forward_pair(imgs[:B], imgs[B:])
torch.cuda.synchronize()
m.kernel_timing(4)
m.forward_pair(imgs[:B], imgs[B:])
torch.cuda.synchronize()
recs = m.kernel_timing_records()
cap = len(recs)
out = (C.c_double * (6 * cap))(); n = C.c_int()
_lib.check(m.lib.sta_kernel_stamps_dump(m._h, cap, out, C.byref(n)))
m.kernel_timing(False)
epi = {0: "f32", 1: "f16", 2: "qkv", 3: "convT", 4: "gelu", 5: "f32r", 6: "head"}
groups = collections.OrderedDict()
for i, (M_, N_, K_, e, a, mx, fam, ms) in enumerate(recs[:n.value]):
    o = [out[6 * i + k] for k in range(6)]
    groups.setdefault((M_, N_, K_, e, a, mx, fam), []).append([ms * 1e3] + o)
print(f"{'M':>8s} {'N':>5s} {'K':>5s} {'epi':>5s} {'A':>4s} {'fam':>3s} {'n':>3s} | {'event':>8s} {'span':>8s} {'WGs':>5s} {'->tile0':>7s} {'loop':>8s} {'epilog':>7s} {'exits':>7s}   (us, means over the launches of the row)")
tot_ev = tot_epi = 0.0
for (M_, N_, K_, e, a, mx, fam), rows in groups.items():
    k = len(rows)
    mean = [sum(r[j] for r in rows) / k for j in range(7)]
    print(f"{M_:8d} {N_:5d} {K_:5d} {epi[e]:>5s} {'conv' if a else 'dns':>4s}{'*' if mx else ' '} {fam:3d} {k:3d} | {mean[0]:8.1f} {mean[2]:8.1f} {int(mean[1]):5d} {mean[3]:7.2f} {mean[4]:8.1f} {mean[5]:7.2f} {mean[6]:7.1f}")
    tot_ev += mean[0] * k; tot_epi += (mean[5] + mean[3]) * k
print(f"sum of event durations {tot_ev / 1e3:.2f} ms; sum over launches of (median prologue + median epilogue) {tot_epi / 1e3:.2f} ms")
