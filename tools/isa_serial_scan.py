"""Static scan of the built library for the store -> load -> s_waitcnt pattern (no GPU).

On gfx9 global loads and stores retire through ONE in-order counter (vmcnt): the data of a load cannot be used before every
store issued ahead of it has been acknowledged.  Written per element (`*c = v + *c`, or a table load between two plane
stores) an epilogue degenerates into one dependent (store-ack, load) round trip per element - round 3 found 16 per 32x32 tile
in the in-place residual epilogue (+4.1 % on the whole step once every load was issued before the first store).

    python tools/isa_serial_scan.py [substring ...]      # kernels whose symbol contains every substring, worst first
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402


def scan():
    """{kernel symbol: {"serial", "ld", "st"}} for every kernel of the built library."""
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(kr.code_object(kr.LIB))
        path = f.name
    txt = subprocess.run([os.path.join(kr.LLVM, "llvm-objdump"), "-d", path], capture_output=True, text=True).stdout
    os.unlink(path)
    cur, stats = None, {}
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = m.group(1)
            stats[cur] = dict(st=0, ld=0, serial=0, pending=False, ld_after=False)
            continue
        t = ln.split()
        if cur is None or not t:
            continue
        op, s = t[0], stats[cur]
        if op.startswith(("global_store", "buffer_store", "global_atomic")):
            s["st"] += 1; s["pending"] = True; s["ld_after"] = False
        elif op.startswith(("global_load", "buffer_load")):
            s["ld"] += 1
            if s["pending"]:
                s["ld_after"] = True
        elif op == "s_waitcnt" and "vmcnt" in ln:
            if s["pending"] and s["ld_after"]:
                s["serial"] += 1
            if "vmcnt(0)" in ln:
                s["pending"] = False; s["ld_after"] = False
    return stats


def main():
    filt = sys.argv[1:]
    stats = scan()
    print("%6s %6s %6s  kernel   (serial = waits on a load issued behind a still-pending store, in program order; static count:\n"
          "                              boundary / fallback paths of a kernel are included whether or not they ever run)" % ("serial", "loads", "stores"))
    for n in sorted(stats, key=lambda n: -stats[n]["serial"]):
        s = stats[n]
        if s["serial"] == 0 or not all(x in n for x in filt):
            continue
        print("%6d %6d %6d  %s" % (s["serial"], s["ld"], s["st"], n[:150]))


if __name__ == "__main__":
    main()
