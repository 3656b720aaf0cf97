"""Dependent-kernel-boundary probe (tools/probes/boundary_probe.hip): standalone process, then the same sweep inside a torch
process next to libsta_mi355.so on torch's streams, then the library's own SLAM-scale chain for comparison.

    python tools/boundary_probe.py [launches]        # -> stdout (tools/final_profiles.sh copies it to profiles/r05_boundary_probe.txt)
"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "tools", "probes", "boundary_probe.hip")
BIN = os.path.join(ROOT, "tools", "probes", "bin")
EXE = os.path.join(BIN, "boundary_probe")
LIB = os.path.join(BIN, "libboundary_probe.so")


def build():
    os.makedirs(BIN, exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-o", EXE, SRC], check=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-DBP_SHARED", "-o", LIB, SRC], check=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 1000
    build()
    if "--build-only" in sys.argv:
        return
    subprocess.run([EXE, str(n)], check=True)
    sys.stdout.flush()
    import torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.sta_frontend import STAFrontend
    m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)          # the library is loaded and has run before the in-process sweep
    img = torch.from_numpy(W.synth_images(1, 224, 224, seed=43, tag=7)).cuda()
    for _ in range(3):
        m._encode_image(img, None, normalize=False)
    torch.cuda.synchronize()
    lib = C.CDLL(LIB)
    lib.bp_sweep.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.bp_sweep(C.c_void_p(torch.cuda.current_stream().cuda_stream), b"torch default stream, inside the torch + libsta_mi355 process", n)
    s = torch.cuda.Stream()
    lib.bp_sweep(C.c_void_p(s.cuda_stream), b"torch.cuda.Stream(), inside the torch + libsta_mi355 process", n)
    # the library's own chain at SLAM scale: one 224x224 encode = 1 gather + 24 x 9 dependent dispatches (+ fills)
    for _ in range(5):
        m._encode_image(img, None, normalize=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        m._encode_image(img, None, normalize=False)
    torch.cuda.synchronize()
    enc = (time.perf_counter() - t0) / 50 * 1e6
    print(f"# library chain: sta_encode 224x224 B=1 = {enc:.1f} us per call; 218 dependent dispatches -> {enc / 218:.2f} us per dispatch "
          f"(work + boundary)")


if __name__ == "__main__":
    main()
