"""MFMA utilisation and HBM bandwidth per kernel from rocprofv3 passes of the SAME bench command:
   * a --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (csv),
   * the FETCH_SIZE / WRITE_SIZE summary written by tools/pmc_summary.py (json),
   * the kernel-trace summary written by tools/rocpd_stats.py (txt, for average durations).

    python tools/pmc_util_summary.py <mfma_pass/*_counter_collection.csv> <traffic.json> <kernel_stats.txt>

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)  (busy cycles are summed over all
SIMDs, GUI_ACTIVE over the 8 XCDs); HBM GB/s = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch / average duration
(FETCH doubled per MI355X_MICROARCH.md's gfx950 note); peaks: 2.5 PFLOP/s dense fp16 MFMA, 8 TB/s HBM3E."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone .*\]", "", name).replace("void ", "")
    return re.sub(r"\(.*\)$", "", name).strip()


def main(pmc_csv, traffic_json, stats_txt):
    acc = defaultdict(lambda: defaultdict(list))
    with open(pmc_csv) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    traffic = json.load(open(traffic_json))
    dur = {}
    for line in open(stats_txt):
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m:
            dur[short(m.group(1).strip())] = (float(m.group(4)), float(m.group(7)))
    print(f"{'kernel':72s} {'pct_time':>8s} {'avg_us':>9s} {'mfma_util':>9s} {'HBM_GB/s':>9s} {'of_8TB/s':>8s}")
    rows = []
    for k, c in acc.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        util = busy / (1024.0 * gui / 8.0) if gui > 0 else 0.0
        key = next((d for d in dur if d[:60] == k[:60]), None)
        avg_us, pct = dur.get(key, (0.0, 0.0))
        tkey = next((t for t in traffic if t[:60] == k[:60]), None)
        gbs = traffic[tkey]["avg_hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9 if tkey and avg_us > 0 else 0.0
        rows.append((pct, k, avg_us, util, gbs))
    for pct, k, avg_us, util, gbs in sorted(rows, reverse=True)[:14]:
        print(f"{k[:72]:72s} {pct:8.2f} {avg_us:9.1f} {util:9.3f} {gbs:9.0f} {gbs / 8000.0:8.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
