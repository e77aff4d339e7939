"""Summarises a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel table (like --stats CSV)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
# optional 2nd argument "kernel_substring:K": only dispatches that start after the K-th launch of that kernel (e.g.
# "adamw_kernel:2" = after two optimizer steps, i.e. without the warm-up steps and the first-call tile autotuning)
t_min = 0
if len(sys.argv) > 2 and ":" in sys.argv[2]:
    kname, kth = sys.argv[2].rsplit(":", 1)
    ends = [r[0] for r in cur.execute("select d.end from %s d join %s s on d.kernel_id=s.id where s.%s like ? order by d.start"
                                      % (disp, sym, namecol), ("%" + kname + "%",))]
    t_min = ends[int(kth) - 1]
q = "select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id where d.start >= %d group by s.%s order by 3 desc" % (namecol, disp, sym, t_min, namecol)
rows = list(cur.execute(q))
total = sum(r[2] for r in rows)
print("%-90s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for name, n, tot, mn, mx in rows:
    name = re.sub(r"\(.*", "", name)[:90]
    print("%-90s %8d %12.3f %10.2f %6.2f%%" % (name, n, tot / 1e6, tot / n / 1e3, 100.0 * tot / total))
print("TOTAL kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
