"""conv3h configuration sweep on the GPU box (experiment switch 0 of sta_debug_set_option): golden check with the kernel
forced on a tiny case, whole-path pairs/s at 8 pairs @512x384 and per-shape conv durations.
    python tools/conv_cfg.py [cfg ...]"""
import collections, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_checks as G
from vista_slam_amd import weights as W, _lib
from vista_slam_amd.sta_frontend import STAFrontend

cfgs = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 8, 9, 10]
for cfg in cfgs:       # correctness first (tiny goldens, conv3h forced, default precision + f16x3)
    for prec in ("f16x3h", "f16x3"):
        m = G.model("tiny", 1.0, prec)
        _lib.check(m.lib.sta_debug_set_option(m._h, 0, cfg))
        r = G.run_golden_case("tiny_48x64_b2", prec, variant=8)
        print(f"cfg {cfg} {prec}: tiny_48x64_b2 max err {max(r.values()):.2e}", flush=True)
        _lib.check(m.lib.sta_debug_set_option(m._h, 0, 0))
G.drop_models()
m = STAFrontend(W.FULL, "cuda:0", precision="f16x3h").load_procedural(seed=43)
B, H, Wd = 8, 384, 512
imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
epi = {0: "f32", 1: "f16", 2: "qkv", 3: "convT", 4: "gelu", 5: "f32r", 6: "head"}
for rep in range(2):
    for cfg in cfgs + [-1]:
        _lib.check(m.lib.sta_set_gemm_variant(m._h, 9 if cfg < 0 else 0))      # -1: the old implicit-GEMM convolutions
        _lib.check(m.lib.sta_debug_set_option(m._h, 0, max(cfg, 0)))
        for _ in range(2):
            m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        line = f"pass {rep} cfg {cfg:2d}: {B / dt:7.2f} pairs/s {dt * 1e3:7.2f} ms |"
        if rep == 1:
            m.kernel_timing(2)
            for _ in range(2):
                m.forward_pair(imgs[:B], imgs[B:])
            torch.cuda.synchronize()
            cap = 4096
            sh = (C.c_int * (6 * cap))(); ms = (C.c_float * cap)(); var = (C.c_int * cap)(); n = C.c_int()
            _lib.check(m.lib.sta_kernel_timing_dump_shapes(m._h, cap, sh, ms, var, C.byref(n)))
            m.kernel_timing(False)
            acc = collections.OrderedDict()
            for i in range(n.value):
                key = tuple(sh[6 * i + q] for q in range(6))
                if key[4] == 1 and key[0] >= 32768:
                    acc.setdefault(key, []).append(ms[i] * 1e3)
            for key, ts in acc.items():
                line += f" {key[0]}x{key[1]}x{key[2]}{'h' if key[3] == 6 else ''}: {sum(ts) / len(ts):6.0f}us x{len(ts) // 2}"
        print(line, flush=True)
