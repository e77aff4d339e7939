"""Timeline of the kernels around the xSlot head in one steady-state training step (rocprofv3 rocpd database of bench.py):
start offset, duration and the idle gap in front of each dispatch, from 12 dispatches before xslot_fwd to 30 after.
usage: python tools_dev/head_section.py <rocpd.db> [before=12] [after=30] [anchor kernel substring = xslot*fwd]
(anchor "adamw": the end of the step -- what the weight-gradient side stream still runs after the compute stream is done)"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
na = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else None)
rows = list(cur.execute("select s.%s, d.start, d.end%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (
    namecol, (", d." + qcol) if qcol else "", disp, sym)))
anchor = sys.argv[4] if len(sys.argv) > 4 else None
idx = [i for i, r in enumerate(rows) if (anchor in r[0] if anchor else ("xslot" in r[0] and "fwd" in r[0]))]
i0 = idx[-2] if len(idx) > 1 else idx[-1]          # the second-to-last step
t0 = rows[i0][1]
prev_end = rows[i0 - nb - 1][2]
print("%-64s %10s %9s %8s %s" % ("kernel", "start_us", "dur_us", "gap_us", "queue"))
for r in rows[i0 - nb:i0 + na]:
    name = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:64]
    print("%-64s %10.1f %9.1f %8.1f %s" % (name, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, r[3] if qcol else ""))
    prev_end = max(prev_end, r[2])
