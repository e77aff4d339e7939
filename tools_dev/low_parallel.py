"""Kernels of the benchmark step that run with FEW workgroups (latency-bound on the critical path): name, launches per
step, workgroups, average microseconds.  usage: python tools_dev/low_parallel.py <rocpd db> [kernel:K after-marker] [max_wgs]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
t_min = 0
if len(sys.argv) > 2 and ":" in sys.argv[2]:
    kname, kth = sys.argv[2].rsplit(":", 1)
    ends = [r[0] for r in cur.execute("select d.end from %s d join %s s on d.kernel_id=s.id where s.%s like ? order by d.start" % (disp, sym, namecol), ("%" + kname + "%",))]
    t_min = ends[int(kth) - 1]
maxw = int(sys.argv[3]) if len(sys.argv) > 3 else 256
gx = [c for c in cols if c.startswith("grid_size")]
wx = [c for c in cols if c.startswith("workgroup_size")]
q = "select s.%s, %s, %s, count(*), sum(d.end-d.start) from %s d join %s s on d.kernel_id=s.id where d.start >= %d group by s.%s, %s, %s" % (
    namecol, ",".join("d." + c for c in gx), ",".join("d." + c for c in wx), disp, sym, t_min, namecol, ",".join("d." + c for c in gx), ",".join("d." + c for c in wx))
rows = []
for r in cur.execute(q):
    name = re.sub(r"\(.*", "", r[0])
    g = r[1:1 + len(gx)]; w = r[1 + len(gx):1 + len(gx) + len(wx)]
    n, tot = r[-2], r[-1]
    wgs = 1
    for a, b in zip(g, w):
        wgs *= max(1, (a + b - 1) // max(b, 1))
    if wgs <= maxw:
        rows.append((tot, name, n, wgs, tot / n / 1e3))
rows.sort(reverse=True)
print("%-72s %6s %6s %9s %10s" % ("kernel", "calls", "WGs", "avg us", "total ms"))
for tot, name, n, wgs, avg in rows[:45]:
    print("%-72s %6d %6d %9.1f %10.3f" % (name[:72], n, wgs, avg, tot / 1e6))
print("total of the listed: %.3f ms" % (sum(r[0] for r in rows) / 1e6))
