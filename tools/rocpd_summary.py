#!/usr/bin/env python3
"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total / average / min / max duration.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, c, tot, avg, mn, mx in rows:
        print(f"{short(n)[:70]:70s} {c:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100.0 * tot / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
