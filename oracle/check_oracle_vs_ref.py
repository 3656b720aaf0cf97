"""Re-generate goldens from the imported reference into a scratch directory and compare them, array by array,
with the committed fixtures under tests/golden/ (THIS CONTAINER ONLY: needs /root/reference).

TEST INFRASTRUCTURE.  The committed `.npz` files are the reference's own outputs on procedural weights and inputs
(`oracle/gen_golden.py`); this script is the proof: it runs the generator again and requires `np.array_equal` on every
array both files hold (a fixture may gain keys - e.g. the `pos_a` / `pos_b` integer positions added in round 4 - but an
array that exists on both sides must not move by one bit).

    python oracle/check_oracle_vs_ref.py                       # a quick default selection (~1 min)
    python oracle/check_oracle_vs_ref.py tiny stress full224   # any case groups of gen_golden.CASES / SEQ_CASES (seq, seqfull)
    python oracle/check_oracle_vs_ref.py --update tiny         # additionally copy fixtures that only GAINED keys
"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def compare(old_path, new_path):
    """-> (n_equal, [keys that differ], [keys only in the new file], [keys only in the old file])"""
    a, b = np.load(old_path), np.load(new_path)
    diff, same = [], 0
    for k in a.files:
        if k not in b.files or k.startswith("meta_"):      # meta_* = the generator's own bookkeeping (case parameters), not reference output
            continue
        x, y = a[k], b[k]
        if x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y, equal_nan=x.dtype.kind == "f"):
            same += 1
        else:
            diff.append(k)
    return same, diff, [k for k in b.files if k not in a.files], [k for k in a.files if k not in b.files]


def main(argv):
    update = "--update" in argv
    groups = [a for a in argv if not a.startswith("--")] or ["tiny", "full224"]
    scratch = tempfile.mkdtemp(prefix="sta_golden_")
    os.environ["STA_GOLDEN_OUT"] = scratch
    sys.path.insert(0, ROOT)
    from oracle import gen_golden as G
    import torch
    torch.set_num_threads(os.cpu_count())
    bad = 0
    for grp in groups:
        for c in (G.CASES.get(grp) or G.SEQ_CASES[grp]):
            G.run_any(c)
            name = c["name"] + ".npz"
            old, new = os.path.join(GOLDEN, name), os.path.join(scratch, name)
            if not os.path.exists(old):
                print(f"[check] {name}: NEW fixture (nothing committed to compare with)")
                if update:
                    shutil.copy(new, old)
                continue
            same, diff, gained, lost = compare(old, new)
            ok = not diff and not lost
            print(f"[check] {name}: {same} arrays bit-identical, {len(diff)} differ {diff}, gained {gained}, lost {lost} -> {'OK' if ok else 'MISMATCH'}")
            if not ok:
                bad += 1
            elif update and gained:
                shutil.copy(new, old)
    shutil.rmtree(scratch, ignore_errors=True)
    print(f"[check] {'all fixtures reproduce bit-exactly' if bad == 0 else str(bad) + ' fixtures do NOT reproduce'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
