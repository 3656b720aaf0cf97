"""Measured tile-family table: per GEMM / convolution shape of the STA forward, the in-model duration
(HIP events around every launch) under the product's family choice and under every forced family, over
B in {1, 2, 4, 8} x {224x224, 384x512}.

    python tools/tile_table.py [steps] > profiles/r03_tile_table.txt

Columns: config | M N K epi A mx | Ho Wo tail | n/step | picked family | us picked | us per forced variant | best | picked/best
(Ho Wo: output geometry of a convolution, 0 0 otherwise; tail: pose-token rows at the end of M that launch_gemm hands to the
skinny tail blocks - the tile rules see M - tail).
Forced variants (sta_set_gemm_variant): 4 = 192x128 wherever N % 128 == 0, 3 = 192x256, 2 = 256x256 (both where N % 256 == 0
and the epilogue is not the RoPE one), 9 = product rules without the halo-tiled 3x3 convolution, 8 = halo-tiled wherever legal,
10 = product rules without the small-grid family, 11 = small-grid family up to 4x its product threshold.
tests/test_tile_table.py parses this file."""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vista_slam_amd import weights as W, _lib  # noqa: E402
from vista_slam_amd import _lib as _hooks_lib; _hooks_lib.use_test_hooks()      # tools use the test-hooks build (include/sta_mi355_debug.h)
from vista_slam_amd.sta_frontend import STAFrontend  # noqa: E402

EPI = {0: "f32", 1: "f16", 2: "qkv", 3: "convT", 4: "gelu", 5: "f32r", 6: "head"}
VARIANTS = [0, 4, 3, 2, 9, 8, 10, 11]


def measure(m, imgs, B, v, steps):
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
    out = collections.OrderedDict()
    for i in range(n.value):
        key = tuple(sh[6 * i + q] for q in range(6))
        out.setdefault(key, []).append((ms[i] * 1e3, var[i]))
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    m = STAFrontend(W.FULL, "cuda:0", precision="f16x3h").load_procedural(seed=43)
    print("# " + __doc__.strip().replace("\n", "\n# "))
    print(f"# steps per cell: {steps}; us = mean launch duration; family ids: 1 = 128x128 register-staged, 2 = 256x256, 3 = 192x256, "
          "5 = 192x128, 6 = small-grid 128x64 (split-K slabs), 7 = paired 192x128 launch, 8 = halo-tiled 3x3")
    print("# config | M N K epi A mx | Ho Wo tail | n/step | picked | us_picked | " + " ".join(f"v{v}" for v in VARIANTS[1:]) + " | best | picked/best")
    for (H, Wd) in ((224, 224), (384, 512)):
        for B in (1, 2, 4, 8):
            imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=0)).cuda()
            per = {v: measure(m, imgs, B, v, steps) for v in VARIANTS}
            _lib.check(m.lib.sta_set_gemm_variant(m._h, 0))
            tot = {v: sum(t for rec in per[v].values() for t, _ in rec) / steps for v in VARIANTS}
            # the product pairs attn.qkv + cross_attn.projk|projv in one launch (family 7: N = 3840 at the full width); the forced
            # variants launch the two GEMMs separately: compare the pair against the sum of its parts
            for key, rec in per[0].items():
                Mr, N, K, e, a, mx = key
                cnt = len(rec) // steps
                Ho = Wo = tail = 0
                if a:                       # every level of the DPT head keeps the input's aspect ratio
                    px = Mr // (2 * B)
                    Ho = int(round((px * H / Wd) ** 0.5)); Wo = px // max(Ho, 1)
                    assert Ho * Wo == px, (Mr, Ho, Wo)
                elif Mr == 2 * B * ((H // 16) * (Wd // 16) + 1):
                    tail = 2 * B
                us0 = sum(t for t, _ in rec) / len(rec)
                fam = rec[0][1]
                cells, alts = [], []
                for v in VARIANTS[1:]:
                    r = per[v].get(key)
                    if r is None and fam == 7:
                        parts = [k for k in per[v] if k[0] == Mr and k[2] == K and k[3] == e and k not in per[0] and k[1] < N]
                        if parts and sum(k[1] for k in parts) == N:
                            us = sum(sum(t for t, _ in per[v][k]) / len(per[v][k]) for k in parts)
                            cells.append(f"{us:.1f}"); alts.append(us)
                            continue
                    if r is None and e in (0, 5):     # in-place residual GEMM: EPI_F32R on the big tiles, EPI_F32 slabs (+ resid_ln_kernel) on the small grid
                        r = per[v].get((Mr, N, K, 5 - e, a, mx))
                    if r is None:
                        cells.append("-")
                        continue
                    us = sum(t for t, _ in r) / len(r)
                    cells.append(f"{us:.1f}[{r[0][1]}]")
                    alts.append(us)
                best = min([us0] + alts)
                print(f"B{B}@{H}x{Wd} | {Mr} {N} {K} {EPI[e]} {'conv' if a else 'dns'} {mx} | {Ho} {Wo} {tail} | {cnt} | {fam} | {us0:.1f} | "
                      + " ".join(cells) + f" | {best:.1f} | {us0 / best:.3f}", flush=True)
            print(f"# B{B}@{H}x{Wd} sum of GEMM/conv launch durations per step (us): " + "  ".join(f"v{v}: {tot[v]:.0f}" for v in VARIANTS), flush=True)


if __name__ == "__main__":
    main()
