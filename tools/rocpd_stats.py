#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats`).
usage: python tools/rocpd_stats.py <results.db> [--md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[1])
    q = (f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for name, n, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-60:]
        print(f"{short:60s} {n:6d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}")


if __name__ == "__main__":
    main()
