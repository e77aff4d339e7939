export TMPDIR=/tmp
O=gpurun_out/r5aa; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/on -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof > $O/on.log 2>&1
DB=$(find $O/on -name "*.db" | head -1)
python tools_dev/gaps.py $DB adamw_kernel:3 > $O/gaps.txt 2>&1
python - $DB > $O/timeline.txt <<'P'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
q = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else "0")
rows = list(cur.execute("select d.start, d.end, s.%s, d.%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (namecol, q, disp, sym)))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
lo, hi = ends[3] + 1, ends[4] + 1
t0 = rows[lo][0]
for s, e, n, qq in rows[lo:hi]:
    print("%9.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qq, re.sub(r"\(.*", "", n)[:90]))
P
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
