"""Run the attention kernel alone (encoder shape by default) - target for rocprofv3 --pmc / timing.
    python tools/attn_one.py [S heads nq nk iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W, _lib
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend
a = [int(x) for x in sys.argv[1:]]
S, heads, nq, nk, iters = (a + [16, 16, 768, 768, 10][len(a):])[:5]
m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
g = torch.Generator().manual_seed(1)
q = torch.randn(S, heads, nq, 64, generator=g).cuda(); k = torch.randn(S, heads, nk, 64, generator=g).cuda()
v = torch.randn(S, heads, nk, 64, generator=g).cuda()
out = torch.empty(S, nq, heads * 64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(iters):
    _lib.check(m.lib.sta_debug_attention(m._h, q.data_ptr(), k.data_ptr(), v.data_ptr(), S, heads, nq, nk, 0, out.data_ptr(), st))
torch.cuda.synchronize()
print("done", S, heads, nq, nk)
