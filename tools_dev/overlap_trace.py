"""What two kernels on two streams do to each other: from a rocprofv3 --kernel-trace database of tools_dev/overlap_probe.py
(or of the training step), for every pair of kernel names (A, B) the time A's dispatches spend overlapped by a B dispatch,
and A's mean duration alone vs while overlapped.  PMC counters cannot show this (rocprofv3 serialises dispatches while it
collects them); the dispatch timestamps can.  usage: python tools_dev/overlap_trace.py <db> <nameA-substring> <nameB-substring>"""
import bisect
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
rows = list(cur.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (namecol, disp, sym)))
A = [(s, e) for n, s, e in rows if sys.argv[2] in n]
B = [(s, e) for n, s, e in rows if sys.argv[3] in n]


def report(X, Y, nx, ny):
    ys = [s for s, e in Y]
    alone, ov = [], []
    for s, e in X:
        inter = 0
        i = max(0, bisect.bisect_left(ys, s) - 4)
        while i < len(Y) and Y[i][0] < e:
            inter += max(0, min(e, Y[i][1]) - max(s, Y[i][0]))
            i += 1
        (ov if inter > 0.05 * (e - s) else alone).append((e - s, inter))
    def mean(v): return sum(v) / max(len(v), 1) / 1e3
    print("%-28s alone: %5d dispatches, mean %8.1f us | overlapped by %-22s %5d dispatches, mean %8.1f us (of which %.1f us "
          "under the other kernel)" % (nx, len(alone), mean([d for d, _ in alone]), ny + ":", len(ov), mean([d for d, _ in ov]),
                                       mean([i for _, i in ov])))


report(A, B, sys.argv[2], sys.argv[3])
report(B, A, sys.argv[3], sys.argv[2])
