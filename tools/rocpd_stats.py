#!/usr/bin/env python
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as markdown."""
import sqlite3
import sys


def main(db, out=None):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["| kernel | calls | total_us | avg_us | % |", "|---|---|---|---|---|"]
    for name, calls, tot, avg, pct in rows:
        lines.append("| `%s` | %d | %.1f | %.3f | %.2f |" % (name[:110], calls, tot, avg, pct))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "a").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
