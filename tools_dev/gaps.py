"""GPU idle time of a rocprofv3 kernel trace: the union of all kernels' busy intervals vs the wall span, per training step,
and the largest gaps with the kernels around them.  usage: python tools_dev/gaps.py <rocpd db> [kernel:K marker]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
rows = list(cur.execute("select d.start, d.end, s.%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (namecol, disp, sym)))
t_min = 0
if len(sys.argv) > 2:
    kname, kth = sys.argv[2].rsplit(":", 1)
    ends = [e for s, e, n in rows if kname in n]
    t_min = ends[int(kth) - 1]
rows = [(s, e, re.sub(r"\(.*", "", n)) for s, e, n in rows if s >= t_min]
steps = sum(1 for r in rows if "adamw_kernel" in r[2])
span = rows[-1][1] - rows[0][0]
busy, cur_end, gaps = 0, rows[0][0], []
prev = rows[0]
for s, e, n in rows:
    if s > cur_end:
        gaps.append((s - cur_end, prev[2], n))
        busy += 0
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
        prev = (s, e, n)
print("steps %d, wall %.3f ms/step, GPU busy (union of kernels) %.3f ms/step, idle %.3f ms/step in %d gaps/step" % (
    steps, span / steps / 1e6, busy / steps / 1e6, (span - busy) / steps / 1e6, len(gaps) // max(steps, 1)))
import collections
by = collections.Counter()
for g, a, b in gaps:
    by[(a[:48], b[:48])] += g
print("largest idle, by (kernel before -> kernel after), microseconds per step:")
for (a, b), g in by.most_common(25):
    print("  %8.1f  %-48s -> %s" % (g / steps / 1e3, a, b))
