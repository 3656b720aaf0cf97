"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total/avg/min/max duration.

    python tools/rocpd_stats.py gpurun_out/prof1/*/*.db > profiles/r01_kernel_stats.txt
    python tools/rocpd_stats.py --tail N gpurun_out/prof1/*/*.db     # only the last N dispatches + the idle time between them
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\[clone .*\]", "", name)
    name = name.replace("void ", "").replace("(AttnParams)", "").replace("(LnParams)", "")
    name = re.sub(r"\(GemmParams(, GemmParams, int)?\)", "", name)
    return name[:110]


def main(paths):
    by_grid = "--by-grid" in paths            # split every kernel by its launch grid (one row per GEMM shape)
    tail = int(paths[paths.index("--tail") + 1]) if "--tail" in paths else 0
    if tail:
        del paths[paths.index("--tail"):paths.index("--tail") + 2]
    paths = [p for p in paths if not p.startswith("--")]
    rows = {}
    if tail:          # timeline of the last `tail` dispatches: busy time, idle gaps (start[i+1] - end[i] > 0), span
        db = sqlite3.connect(paths[0])
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
        ev = sorted(db.execute(f"select start, end, {namecol} from kernels"))[-tail:]
        busy = sum(e - s for s, e, _ in ev) / 1e3
        gaps = [max(0, ev[i + 1][0] - ev[i][1]) / 1e3 for i in range(len(ev) - 1)]
        span = (max(e for _, e, _ in ev) - ev[0][0]) / 1e3
        # several streams: kernels overlap - the time at least one kernel is running (union of the intervals) and the average
        # number of kernels in flight while the GPU is busy
        union, cur_s, cur_e = 0, ev[0][0], ev[0][1]
        for s_, e_, _ in ev[1:]:
            if s_ > cur_e:
                union += cur_e - cur_s; cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        union = (union + cur_e - cur_s) / 1e3
        print(f"last {len(ev)} dispatches: span {span:.1f} us, sum of kernel durations {busy:.1f} us, idle between kernels {sum(gaps):.1f} us "
              f"({len([g for g in gaps if g > 0])} gaps, median {sorted(gaps)[len(gaps) // 2]:.2f} us, max {max(gaps):.1f} us); "
              f"at least one kernel running {union:.1f} us ({100 * union / span:.1f} % of the span), {busy / union:.2f} kernels in flight on average while busy")
        for s_, e_, n_ in ev:
            r = rows.setdefault(short(n_)[:88], [0, 0.0, 1e30, 0.0]); d = (e_ - s_) / 1e3
            r[0] += 1; r[1] += d; r[2] = min(r[2], d); r[3] = max(r[3], d)
        paths = []
    for p in paths:
        db = sqlite3.connect(p)
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
        gcol = next((c for c in cols if "grid" in c.lower() and c.lower().endswith("x")), None)
        wcol = next((c for c in cols if "workgroup" in c.lower() and c.lower().endswith("x")), None)
        sel = f"select {namecol}, start, end" + (f", {gcol}" if by_grid and gcol else ", 0") + (f", {wcol}" if by_grid and wcol else ", 1") + " from kernels"
        for name, start, end, g, w in db.execute(sel):
            d = (end - start) / 1e3
            key = short(name)[:88] + (f" grid={int(g) // max(int(w), 1)}" if by_grid else "")
            r = rows.setdefault(key, [0, 0.0, 1e30, 0.0])
            r[0] += 1; r[1] += d; r[2] = min(r[2], d); r[3] = max(r[3], d)
    tot = sum(r[1] for r in rows.values())
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:110s} {r[0]:7d} {r[1] / 1e3:10.3f} {r[1] / r[0]:10.2f} {r[2]:9.2f} {r[3]:9.2f} {100 * r[1] / tot:6.2f}")
    print(f"{'TOTAL':110s} {sum(r[0] for r in rows.values()):7d} {tot / 1e3:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1:])
