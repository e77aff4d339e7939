"""Per-kernel sums of every PMC counter in a rocprofv3 rocpd database.  usage: python tools_dev/pmc_dump.py <db> [kernel_substring]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
ev = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = ("select s.%s, i.name, count(distinct d.id), sum(e.value) from %s d join %s s on d.kernel_id=s.id join %s e on e.event_id=d.event_id "
     "join %s i on i.id=e.pmc_id where s.%s like ? group by s.%s, i.name" % (namecol, disp, sym, ev, info, namecol, namecol))
for name, cname, n, v in cur.execute(q, ("%" + flt + "%",)):
    print("%-60s %-32s launches %4d  per launch %.4g" % (re.sub(r"\(.*", "", name)[:60], cname, n, v / n))
