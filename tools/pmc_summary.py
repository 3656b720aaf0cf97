"""Summarise rocprofv3 --pmc CSV passes (FETCH_SIZE / WRITE_SIZE, collected in SEPARATE runs) per kernel.

    python tools/pmc_summary.py gpurun_out/pmc_fetch/*/*_counter_collection.csv \
                                gpurun_out/pmc_write/*/*_counter_collection.csv > profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read (MI355X_MICROARCH.md, "HBM"), so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone .*\]", "", name).replace("void ", "")
    return re.sub(r"\(.*\)$", "", name).strip()


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, c in acc.items():
        n = max(len(v) for v in c.values())
        fetch = sum(c.get("FETCH_SIZE", [])) / max(len(c.get("FETCH_SIZE", [])), 1)
        write = sum(c.get("WRITE_SIZE", [])) / max(len(c.get("WRITE_SIZE", [])), 1)
        out[k] = {"launches": n, "avg_FETCH_SIZE_KiB": round(fetch, 1), "avg_WRITE_SIZE_KiB": round(write, 1),
                  "avg_hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
    print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["avg_hbm_bytes_per_launch"] * kv[1]["launches"])), indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
