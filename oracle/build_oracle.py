"""Build the CPU oracle ops library (gcc, OpenMP).  TEST INFRASTRUCTURE.

    python oracle/build_oracle.py   # -> oracle/libsta_oracle.so
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "sta_oracle_ops.c")
LIB = os.path.join(HERE, "libsta_oracle.so")


def build(force=False, verbose=True):
    import hashlib
    with open(SRC, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    hfile = LIB + ".srchash"
    if not force and os.path.exists(LIB) and os.path.exists(hfile) and open(hfile).read().strip() == digest:
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: cannot build the CPU oracle")
    # -march=native would bake the build host's ISA into a .so that travels to the GPU box: keep it
    # portable (x86-64-v3 = AVX2+FMA; falls back to generic if the compiler rejects it).
    for arch in ("-march=x86-64-v3", "-mavx2 -mfma", ""):
        cmd = [gcc, "-O3", "-fopenmp", "-fPIC", "-shared", "-fno-math-errno"] + arch.split() + ["-o", LIB, SRC, "-lm"]
        if verbose:
            print("[oracle build]", " ".join(cmd), flush=True)
        if subprocess.run(cmd).returncode == 0:
            with open(hfile, "w") as f:
                f.write(digest)
            return LIB
    raise RuntimeError("oracle build failed")


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
