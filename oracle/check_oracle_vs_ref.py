"""Re-generate goldens from the imported reference into a scratch directory and compare them, array by array,
with the committed fixtures under tests/golden/ (THIS CONTAINER ONLY: needs /root/reference).

TEST INFRASTRUCTURE.  The committed `.npz` files are the reference's own outputs on procedural weights and inputs
(`oracle/gen_golden.py`); this script is the proof: it runs the generator again and requires `np.array_equal` on every
array both files hold (a fixture may gain keys - e.g. the `pos_a` / `pos_b` integer positions added in round 4 - but an
array that exists on both sides must not move by one bit).

    python oracle/check_oracle_vs_ref.py                       # a quick default selection (~1 min)
    python oracle/check_oracle_vs_ref.py tiny stress full224   # any case groups of gen_golden.CASES / SEQ_CASES (seq, seqfull)
    python oracle/check_oracle_vs_ref.py --update tiny         # additionally copy fixtures that only GAINED keys
    python oracle/check_oracle_vs_ref.py ckpt                  # the real-checkpoint kit on a STAND-IN file (below; ~10 min)

`ckpt`: torch.save({'model': procedural full-architecture weights}) -> `gen_golden.gen_checkpoint(file)` (what
`python oracle/gen_golden.py --checkpoint FILE` runs) -> the three fixtures it writes must be bit-identical, array by array, to the
committed fixtures of the same weights handed over directly: ckpt_224_b1 == full_224_b1, ckpt_384x512_b1 == full_384x512_b1,
seq_tum_ckpt_224 == seq_tum_full_224_t075.  The file path of the kit is proven on everything but the real download.
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


def check_checkpoint_kit(scratch):
    """The real-checkpoint acceptance kit on a stand-in checkpoint file holding the procedural weights."""
    import torch
    from oracle import gen_golden as G
    from vista_slam_amd import weights as W
    sd = W.state_dict(W.FULL, seed=43)
    path = os.path.join(scratch, "standin_frontend_sta_weights.pth")
    torch.save({"model": {k: torch.from_numpy(v.copy()) for k, v in sd.items()}, "epoch": 0}, path)
    fp = W.state_dict_fingerprint(sd)
    del sd
    print(f"[check] stand-in checkpoint {path}: {os.path.getsize(path) / 1e9:.2f} GB, fingerprint {fp[:16]}...", flush=True)
    G.gen_checkpoint(path, tag="ckpt", cfg=W.FULL)
    os.remove(path)
    bad = 0
    for new, old in (("ckpt_224_b1", "full_224_b1"), ("ckpt_384x512_b1", "full_384x512_b1"), ("seq_tum_ckpt_224", "seq_tum_full_224_t075")):
        same, diff, gained, lost = compare(os.path.join(GOLDEN, old + ".npz"), os.path.join(scratch, new + ".npz"))
        z = np.load(os.path.join(scratch, new + ".npz"))
        ok = not diff and not lost and str(z["ckpt_fingerprint"]) == fp
        print(f"[check] {new}.npz (from the FILE) vs committed {old}.npz: {same} arrays bit-identical, {len(diff)} differ {diff}, "
              f"gained {[k for k in gained if not k.startswith('range_')][:6]} (+{sum(k.startswith('range_') for k in gained)} range_* arrays), lost {lost}, "
              f"fingerprint {'matches' if str(z['ckpt_fingerprint']) == fp else 'DIFFERS'} -> {'OK' if ok else 'MISMATCH'}")
        bad += 0 if ok else 1
    print(open(os.path.join(scratch, "ckpt_ranges.txt")).read())
    return bad


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
        if grp == "ckpt":
            G.OUT = scratch
            bad += check_checkpoint_kit(scratch)
            continue
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
