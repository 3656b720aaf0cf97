"""Build the gfx950 C-ABI libraries in-tree with hipcc (no torch, no cmake).

    python -m vista_slam_amd.build          # -> vista_slam_amd/libsta_mi355.so        the product: exports include/sta_mi355.h only
                                            #    vista_slam_amd/libsta_mi355_test.so   the same translation unit + -DSTA_TEST_HOOKS:
                                            #    additionally the kernel-level test / micro-benchmark entry points of
                                            #    include/sta_mi355_debug.h (loaded by tests/ and tools/ only)
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libsta_mi355.so")
TEST_LIB = os.path.join(PKG, "libsta_mi355_test.so")
SOURCES = ["sta_api.hip"]
DEPS = ["sta_exports.map", "sta_api.hip", "sta_launch.inc", "sta_forward.inc", "sta_debug.inc", "sta_rows.inc", "sta_bench.inc", "gemm.h", "gemm2.h", "conv3h.h", "attention.h", "elementwise.h", "sta_common.h",
        os.path.join("..", "..", "include", "sta_mi355.h"), os.path.join("..", "..", "include", "sta_mi355_debug.h")]


def hipcc_path():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


# -fvisibility=hidden: the dynamic symbol table is the STA_API declarations of include/*.h and nothing else (no __device_stub__
# launch stubs, no helpers); tests/test_cabi_symbols.py asserts it
# + a linker version script (csrc/sta_exports.map) for what visibility attributes cannot reach: hipcc's kernel handle objects,
# libstdc++ template instantiations, the __hip_cuid_ marker
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-value",
              "-Wl,--version-script=" + os.path.join(CSRC, "sta_exports.map")]
HASH_FILE = LIB + ".srchash"
TEST_HASH_FILE = TEST_LIB + ".srchash"


def extra_flags():
    """STA_DEV_FAST=1: development build without the precision-f16 kernel forms (half the compile time; the f16 tests fail
    loudly on it).  STA_BENCH_EXPERIMENTS=1: the retired tile shapes / main-loop ablations of tools/gemm_tiles.py."""
    f = []
    if os.environ.get("STA_DEV_FAST") == "1":
        f.append("-DSTA_DEV_FAST")
    if os.environ.get("STA_BENCH_EXPERIMENTS") == "1":
        f.append("-DSTA_BENCH_EXPERIMENTS")
    return f


def source_hash():
    """Content hash of every source the library is built from (mtimes do not survive the gpurun
    snapshot copy, so staleness is decided by content) and of the build flags."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join([f.replace(CSRC, "csrc") for f in BASE_FLAGS] + extra_flags()).encode())      # (path-independent: the snapshot on the GPU box lives elsewhere)
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale(lib=LIB, hash_file=HASH_FILE):
    if not os.path.exists(lib) or not os.path.exists(hash_file):
        return True
    with open(hash_file) as f:
        return f.read().strip() != source_hash()


def _cmd(hipcc, out, hooks):
    return ([hipcc] + BASE_FLAGS + ["-o", out] + extra_flags() +
            (["-DSTA_TEST_HOOKS"] if hooks else []) + [os.path.join(CSRC, s) for s in SOURCES])


def build_lib(force=False, verbose=True, test_hooks=True):
    """Build the product library and (test_hooks) the test-hooks library; both compile concurrently (one hipcc process each,
    ~100 s).  Returns the product library's path."""
    jobs = []
    if force or is_stale(LIB, HASH_FILE):
        jobs.append((LIB, HASH_FILE, False))
    if test_hooks and (force or is_stale(TEST_LIB, TEST_HASH_FILE)):
        jobs.append((TEST_LIB, TEST_HASH_FILE, True))
    if not jobs:
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libsta_mi355.so (ROCm toolchain required)")
    procs = []
    for out, _hf, hooks in jobs:
        cmd = _cmd(hipcc, out, hooks)
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd, cwd=CSRC), cmd))
    failed = None
    for (p, cmd), (out, hf, _hooks) in zip(procs, jobs):
        if failed is not None:             # a sibling compile already failed: do not leave this one running un-waited
            p.kill()
            p.wait()
            continue
        if p.wait() != 0:
            failed = subprocess.CalledProcessError(p.returncode, cmd)
            continue
        with open(hf, "w") as f:
            f.write(source_hash())
    if failed is not None:
        raise failed
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
