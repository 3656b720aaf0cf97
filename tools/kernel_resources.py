"""Per-kernel register / LDS / scratch usage of the built library (reads the code object's metadata notes; no GPU).

    python tools/kernel_resources.py [substring ...]      # rows whose demangled name contains every substring
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vista_slam_amd", "libsta_mi355.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(lib):
    blob = open(lib, "rb").read()
    # the fat binary holds one ELF for gfx950 after the __CLANG_OFFLOAD_BUNDLE__ header; find the AMDGPU ELF by e_machine 224
    out = []
    for m in re.finditer(b"\x7fELF", blob):
        o = m.start()
        if blob[o + 18:o + 20] == (224).to_bytes(2, "little"):
            out.append(o)
    assert out, "no AMDGPU code object found"
    return blob[out[0]:]


def main():
    filt = sys.argv[1:]
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(code_object(LIB))
        path = f.name
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
    os.unlink(path)
    rows = []
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, blk)
            return m.group(1) if m else "?"
        name = g("name")
        try:
            name = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip()
        except OSError:
            pass
        rows.append((name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
    print("%-110s %5s %5s %5s %8s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "spill"))
    for r in sorted(rows):
        if all(s in r[0] for s in filt):
            print("%-110s %5s %5s %5s %8s %8s %6s" % (r[0][:110], r[1], r[2], r[3], r[4], r[5], r[6]))


if __name__ == "__main__":
    main()
