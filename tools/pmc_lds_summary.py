"""LDS utilisation per kernel from one rocprofv3 --pmc pass (SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE) of tools/model_steps.py:

    python tools/pmc_lds_summary.py <pass/*_counter_collection.csv> [kernel_stats.txt]

lds_util = SQ_LDS_IDX_ACTIVE summed over the CUs / (256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)  - the fraction of the launch's cycles a
CU's LDS spends on indexed operations (rocprofv3's derived `LdsUtil`); conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
Round 5: evidence for 'the f16mx DPT kernels are LDS-fragment-read bound' (DESIGN.md section 8)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone .*\]", "", name).replace("void ", "")
    return re.sub(r"\(.*\)$", "", name).strip()


def main(pmc_csv, stats_txt=None):
    acc = defaultdict(lambda: defaultdict(list))
    with open(pmc_csv) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    pct = {}
    if stats_txt:
        for line in open(stats_txt):
            m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
            if m:
                pct[short(m.group(1).strip())[:60]] = (float(m.group(7)), float(m.group(4)))
    rows = []
    for k, c in acc.items():
        if "SQ_LDS_IDX_ACTIVE" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        act = sum(c["SQ_LDS_IDX_ACTIVE"]) / len(c["SQ_LDS_IDX_ACTIVE"])
        gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        conf = sum(c.get("SQ_LDS_BANK_CONFLICT", [0.0])) / max(1, len(c.get("SQ_LDS_BANK_CONFLICT", [0.0])))
        util = act / (256.0 * gui / 8.0) if gui > 0 else 0.0
        p, us = pct.get(k[:60], (0.0, 0.0))
        rows.append((p, k, us, util, conf / act if act > 0 else 0.0))
    print(f"{'kernel':72s} {'pct_time':>8s} {'avg_us':>9s} {'lds_util':>9s} {'conflict':>9s}")
    for p, k, us, util, cf in (sorted(rows, reverse=True)[:16] if pct else sorted(rows, key=lambda r: -r[3])):      # without a stats file: every kernel, by LDS utilisation
        print(f"{k[:72]:72s} {p:8.2f} {us:9.1f} {util:9.3f} {cf:9.4f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
