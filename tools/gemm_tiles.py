"""GEMM tile-family experiments on the GPU box (replaces the one-off gemm_bench*.py scripts).

    python tools/gemm_tiles.py micro [tile ids ...]     # product shapes x tile families, back-to-back launches
    python tools/gemm_tiles.py model [variant ids ...]  # whole forward (8 pairs @512x384) with a forced tile family
    python tools/gemm_tiles.py ablate TILE              # component ablations of one tile family (2 or 6)

Tile ids of sta_bench_gemm (csrc/sta_bench.inc): 1 = 128x128 register-staged, 2 = 256x256/8 waves, 3 = 256x128,
5 = 192x256, 6 = 192x128/8 waves (product), 20 = 192x128/6 waves, 21 = 256x256/16 waves, 22 = 192x256/12 waves,
23 = 128x256, 25 = 192x96/6 waves, 26 = 384x128; +100 = the same tile with the in-place residual epilogue.
Variant ids of sta_set_gemm_variant: 0 = product selection, 2..4 = forced families, 7 / 8 / 9 = experiments.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vista_slam_amd import weights as W, _lib  # noqa: E402
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend  # noqa: E402

SHAPES = [("enc qkv", 12288, 3072, 1024), ("enc proj", 12288, 1024, 1024), ("enc fc1", 12288, 4096, 1024),
          ("enc fc2", 12288, 1024, 4096), ("dec qkv", 12304, 2304, 768), ("dec fc1", 12304, 3072, 768),
          ("dec fc2", 12304, 768, 3072), ("dec proj", 12304, 768, 768), ("dec ckv", 12304, 1536, 768),
          ("sq 8192", 8192, 8192, 8192)]


def micro(tiles):
    m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
    m.bench_gemm(12288, 4096, 1024, iters=5, tile=6)      # warm-up (clocks, code objects)
    print("algorithmic TFLOP/s (f16x3: x3 issued MFMA products) [effective GHz]")
    print(f"{'shape':9s} {'MxNxK':>18s} " + " ".join(f"{'t' + str(t):>14s}" for t in tiles))
    for name, M, N, K in SHAPES:
        row = f"{name:9s} {M:6d}x{N:5d}x{K:5d} "
        for t in tiles:
            try:
                ms = m.bench_gemm(M, N, K, iters=10, tile=t)
                ghz = m.lib.sta_bench_gemm_last_ghz()
                row += f" {2.0 * M * N * K / ms / 1e9:7.1f}[{ghz:4.2f}]"
            except Exception as e:   # noqa: BLE001
                row += f" {'err':>13s}"
                print("   ", e)
        print(row, flush=True)


def model(variants, steps=6):
    m = STAFrontend(W.FULL, "cuda:0", precision="f16x3").load_procedural(seed=43)
    B, H, Wd = 8, 384, 512
    imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
    ref = None
    for rep in range(2):           # two interleaved passes: box drift shows up as a difference between them
        for v in variants:
            _lib.check(m.lib.sta_set_gemm_variant(m._h, v))
            for _ in range(2):
                out = m.forward_pair(imgs[:B], imgs[B:])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = m.forward_pair(imgs[:B], imgs[B:])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            pts = out[0]["pts3d_pred"].float()
            if ref is None:
                ref = pts.clone()
            err = float((pts - ref).norm() / ref.norm())
            print(f"pass {rep} variant {v}: {B / dt:7.2f} pairs/s  {dt * 1e3:7.2f} ms/step  rel diff vs first {err:.1e}", flush=True)


def shapes(variants, steps=3):
    """Per-shape in-model duration (HIP events around every GEMM / conv launch) for each forced tile family."""
    import collections
    import ctypes as C
    m = STAFrontend(W.FULL, "cuda:0", precision=os.environ.get("STA_PRECISION", "f16x3h")).load_procedural(seed=43)
    B, H, Wd = int(os.environ.get("AB_B", 8)), 384, 512
    for kv in os.environ.get("STA_DEBUG_OPT", "").split(","):          # e.g. STA_DEBUG_OPT=3:1 -> sta_debug_set_option(h, 3, 1)
        if ":" in kv:
            _lib.check(m.lib.sta_debug_set_option(m._h, int(kv.split(":")[0]), int(kv.split(":")[1])))
    imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
    table = collections.OrderedDict()
    for v in variants:
        _lib.check(m.lib.sta_set_gemm_variant(m._h, v))
        for _ in range(2):
            m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        m.kernel_timing(2)
        for _ in range(steps):
            m.forward_pair(imgs[:B], imgs[B:])
        torch.cuda.synchronize()
        cap = 8192
        sh = (C.c_int * (6 * cap))(); ms = (C.c_float * cap)(); var = (C.c_int * cap)(); n = C.c_int()
        _lib.check(m.lib.sta_kernel_timing_dump_shapes(m._h, cap, sh, ms, var, C.byref(n)))
        m.kernel_timing(False)
        for i in range(n.value):
            key = tuple(sh[6 * i + q] for q in range(6))
            table.setdefault(key, collections.defaultdict(list))[v].append((ms[i] * 1e3, var[i]))
    epi = {0: "f32", 1: "f16", 2: "qkv", 3: "convT", 4: "gelu", 5: "f32r", 6: "head"}
    print("(* = f16mx arithmetic: 2 MFMA units per algorithmic FLOP instead of 3)")
    print(f"{'M':>8s} {'N':>5s} {'K':>5s} {'epi':>5s} {'A':>4s}  {'n/step':>5s} {'GF':>8s} | " + " | ".join(f"v{v}: us (TF) [family]" for v in variants))
    tot = {v: 0.0 for v in variants}
    for key, per in table.items():
        M, N, K, e, a, mx = key
        gf = 2.0 * M * N * K / 1e9
        cnt = max(len(per[v]) for v in variants) // steps
        row = f"{M:8d} {N:5d} {K:5d} {epi[e]:>5s} {'conv' if a else 'dns':>4s}{'*' if mx else ' '} {cnt:5d} {gf:8.2f} |"
        for v in variants:
            ts = [t for t, _ in per[v]]
            if not ts:
                row += "        -              |"
                continue
            avg = sum(ts) / len(ts)
            tot[v] += sum(ts) / steps
            row += f" {avg:8.1f} ({gf / avg * 1e3:5.0f}) [{per[v][0][1]}] |"
        print(row, flush=True)
    print("sum of GEMM/conv launch durations per step (ms): " + "  ".join(f"v{v}: {tot[v] / 1e3:.2f}" for v in variants))


def ablate(tile):
    m = STAFrontend(W.TINY, "cuda:0", precision="f16x3").load_procedural()
    names = {0: "full", 1: "no DMA", 2: "no LDS reads", 3: "MFMA only", 4: "no MFMA", 5: "LDS reads only", 6: "DMA only", 7: "barriers only"}
    for name, M, N, K in (SHAPES[2], SHAPES[3], SHAPES[-1]):
        row = f"{name:9s}"
        for abl, an in names.items():
            ms = m.bench_gemm(M, N, K, iters=10, tile=tile, ablation=abl)
            row += f"  {an}: {ms * 1e3:7.1f}us"
        print(row, flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "micro"
    ids = [int(x) for x in sys.argv[2:]]
    if mode == "micro":
        micro(ids or [6, 20, 2, 21, 22, 23, 26, 25])
    elif mode == "model":
        model(ids or [0, 7, 8, 9])
    elif mode == "shapes":
        shapes(ids or [0, 8, 9])
    else:
        ablate(ids[0] if ids else 6)
