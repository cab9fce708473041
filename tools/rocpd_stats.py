#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) as a per-kernel stats table."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
        "group by name order by 6 desc"))
    tot = sum(r[5] for r in rows) or 1
    lines = ["%-70s %8s %12s %12s %12s %12s %7s" % ("Name", "Calls", "TotalDur(ns)", "Avg(ns)", "Min(ns)", "Max(ns)", "Pct")]
    for r in rows:
        lines.append("%-70s %8d %12d %12.1f %12d %12d %6.2f%%" % (r[0][:70], r[1], r[5], r[2], r[3], r[4], 100.0 * r[5] / tot))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
