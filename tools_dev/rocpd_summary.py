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
q = "select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.%s order by 3 desc" % (namecol, disp, sym, namecol)
rows = list(cur.execute(q))
total = sum(r[2] for r in rows)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
print("%-90s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for name, n, tot, mn, mx in rows:
    name = re.sub(r"\(.*", "", name)[:90]
    print("%-90s %8d %12.3f %10.2f %6.2f%%" % (name, n, tot / 1e6, tot / n / 1e3, 100.0 * tot / total))
print("TOTAL kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
