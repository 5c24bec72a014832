#!/usr/bin/env python3
"""Kernel-trace excerpt showing the two tile groups' launches overlapping in time (the pairing behind `frac_paired`).
From a rocprofv3 --kernel-trace rocpd database of `bench.py --steps 1 --warmup 0`: the dispatches of a window of the LAST solve, per
hardware queue (one per CU-masked stream), with start / end relative to the window's first start, plus the overlap statistics of the
whole solve.  usage: group_overlap.py results.db out.md [first_iteration] [iterations]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    dcol = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in dcol else ("stream_id" if "stream_id" in dcol else None)
    rows = c.execute("select s.%s, d.start, d.end, %s from %s d join %s s on d.kernel_id = s.id order by d.start" %
                     (namecol, ("d." + qcol) if qcol else "0", kd, ks)).fetchall()
    rows = [(n.replace("cddp_dev::", "").split("<")[0].replace("void ", ""), a, b, q) for n, a, b, q in rows]
    # the last solve: from the last pair of k_init dispatches on
    inits = [i for i, r in enumerate(rows) if r[0].startswith("k_init")]
    first = inits[-2] if len(inits) >= 2 and rows[inits[-1]][3] != rows[inits[-2]][3] else inits[-1]
    solve = rows[first:]
    queues = sorted({r[3] for r in solve if r[0].startswith("k_forward")})
    it0 = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    nit = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    # whole-solve overlap of the queues' busy intervals
    busy = {q: sorted((a, b) for n, a, b, qq in solve if qq == q) for q in queues}
    def total(iv):
        return sum(b - a for a, b in iv)
    def inter(x, y):
        i = j = 0; s = 0
        while i < len(x) and j < len(y):
            lo = max(x[i][0], y[j][0]); hi = min(x[i][1], y[j][1])
            if hi > lo: s += hi - lo
            if x[i][1] < y[j][1]: i += 1
            else: j += 1
        return s
    lines = ["# Tile groups on CU-masked streams: launches of both groups in time (rocprofv3 --kernel-trace)", "",
             "Hardware queues carrying rollout launches in the last solve: %s" % queues, ""]
    if len(queues) >= 2:
        a, b = busy[queues[0]], busy[queues[1]]
        span = max(x[1] for x in a + b) - min(x[0] for x in a + b)
        lines += ["| | ms |", "|---|---|", "| solve span (first start to last end) | %.3f |" % (span / 1e6),
                  "| kernel time, queue %s | %.3f |" % (queues[0], total(a) / 1e6), "| kernel time, queue %s | %.3f |" % (queues[1], total(b) / 1e6),
                  "| both queues busy at the same time | %.3f |" % (inter(a, b) / 1e6),
                  "| mean concurrency (kernel time of both / span) | %.2f |" % ((total(a) + total(b)) / span), ""]
        fa = [(x, y) for n, x, y, q in solve if q == queues[0] and n.startswith("k_forward")]
        fb = [(x, y) for n, x, y, q in solve if q == queues[1] and n.startswith("k_forward")]
        lines += ["Rollout launches (`k_forward_ipddp_pc`): %.3f ms on queue %s, %.3f ms on queue %s, %.3f ms of them at the same time (%.0f %% of the shorter)." %
                  (total(fa) / 1e6, queues[0], total(fb) / 1e6, queues[1], inter(fa, fb) / 1e6, 100.0 * inter(fa, fb) / max(1, min(total(fa), total(fb)))), ""]
    # excerpt: iterations it0 .. it0 + nit - 1 counted by sweep launches per queue
    lines += ["## Excerpt: outer iterations %d - %d of the last solve (us after the excerpt's first start)" % (it0, it0 + nit - 1), "",
              "| queue | kernel | start | end | us |", "|---|---|---|---|---|"]
    ex = []
    for q in queues:
        nsw = 0
        for n, a, b, qq in solve:
            if qq != q: continue
            if n.startswith("k_backward"): nsw += 1
            # an iteration starts with its first kernel: the role-split sweep (or k_condense / k_derivs when they are launches)
            if n.startswith(("k_condense", "k_derivs")) : pass
            if it0 <= nsw < it0 + nit or (nsw == it0 + nit and not n.startswith("k_backward") and False):
                ex.append((a, b, q, n))
    if ex:
        z = min(a for a, _, _, _ in ex)
        for a, b, q, n in sorted(ex):
            lines.append("| %s | `%s` | %.1f | %.1f | %.1f |" % (q, n, (a - z) / 1e3, (b - z) / 1e3, (b - a) / 1e3))
    out = "\n".join(lines)
    open(sys.argv[2], "w").write(out + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main()
