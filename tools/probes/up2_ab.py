import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, numpy as np
from vista_slam_amd import _lib, weights as W
from vista_slam_amd.sta_frontend import STAFrontend
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
outs = {}
for tag, path in (("new", None), ("old", os.path.join(R, "tools", "ab", "libsta_old.so"))):
    prod = _lib._lib if _lib._lib is not None else _lib.load()
    if path: _lib._lib = _lib.load_other(path)
    m = STAFrontend(W.TINY, "cuda:0", precision="f16x3h").load_procedural()
    _lib.check(m.lib.sta_debug_set_option(m._h, 4, 1))
    g = torch.Generator().manual_seed(5)
    n, H, Wd, Cd = 2, 24, 40, 128
    x = torch.randn(n, H, Wd, Cd, generator=g).cuda()
    out = torch.empty(n, 2 * H, 2 * Wd, Cd, device="cuda")
    _lib.check(m.lib.sta_debug_up2(m._h, x.data_ptr(), n, H, Wd, Cd, 2 * H, 2 * Wd, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    outs[tag] = out.cpu().numpy().copy()
    ref = torch.nn.functional.interpolate(x.cpu().permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1).numpy()
    print(tag, "rel l2 vs torch", np.sqrt(((outs[tag] - ref) ** 2).sum() / (ref ** 2).sum()))
    _lib._lib = prod
d = outs["new"] != outs["old"]
print("differing elements", d.sum(), "of", d.size, "per channel%8:", [int(d[..., k::8].sum()) for k in range(8)])
