#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into a per-kernel table
(calls, total ms, avg us, min, max, % of GPU kernel time) -- the same figures as rocprofv3's
kernel_stats.csv.  usage: summarize_rocpd.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    rows = c.execute("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                     "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, kd, ks, namecol)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, cnt, s, a, mn, mx in rows:
        n = n.replace("cddp_dev::", "")
        if len(n) > 110:
            n = n[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (n, cnt, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
