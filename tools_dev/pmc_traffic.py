"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over the timed steps of bench.py.
usage: python tools_dev/pmc_traffic.py <fetch.db> <write.db> [kernel_substring:K] [out.json] [steps]   (see rocpd_summary.py)
FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE reports half of the bytes of wide streaming reads
(MI355X_MICROARCH.md, HBM section), so it is doubled."""
import json
import re
import sqlite3
import sys


def per_kernel(path, after):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    ev = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    namecol = "display_name" if "display_name" in scols else "kernel_name"
    t_min = 0
    if after:
        kname, kth = after.rsplit(":", 1)
        ends = [r[0] for r in cur.execute("select d.end from %s d join %s s on d.kernel_id=s.id where s.%s like ? order by d.start"
                                          % (disp, sym, namecol), ("%" + kname + "%",))]
        t_min = ends[int(kth) - 1]
    q = ("select s.%s, count(*), sum(e.value) from %s d join %s s on d.kernel_id=s.id join %s e on e.event_id=d.event_id "
         "where d.start >= %d group by s.%s" % (namecol, disp, sym, ev, t_min, namecol))
    return {re.sub(r"\(.*", "", n): (c, v) for n, c, v in cur.execute(q)}


fetch = per_kernel(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 else None)
write = per_kernel(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
rows = []
for name, (n, kb) in fetch.items():
    wn, wkb = write.get(name, (n, 0.0))
    rd = 2.0 * kb * 1024 / n            # gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes
    wr = wkb * 1024 / max(wn, 1)
    rows.append((rd + wr, name, n, rd, wr))
rows.sort(reverse=True)
print("%-70s %7s %14s %14s %14s" % ("kernel", "calls", "read MB/launch", "write MB/launch", "total MB/launch"))
out = {}
for tot, name, n, rd, wr in rows[:60]:
    print("%-70s %7d %14.2f %14.2f %14.2f" % (name[:70], n, rd / 1e6, wr / 1e6, tot / 1e6))
for tot, name, n, rd, wr in rows:
    out[name] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
all_bytes = sum(tot * n for tot, name, n, rd, wr in rows)
print("all kernels: %.1f MB per step (%d profiled steps)" % (all_bytes / steps / 1e6, steps))
import hashlib, os
def _table_sha16():
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scouter_amd", "tuning", "gfx950.json")
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
json.dump({"command": "SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} -- python bench.py "
                      "--steps 3 --warmup 2 --no-cpu-baseline --no-prof (dispatches after the warm-up only)",
           "correction": "read = 2 x FETCH_SIZE KB (gfx950), write = WRITE_SIZE KB", "tuning_sha16": _table_sha16(),
           "steps": steps, "hbm_bytes_per_step": all_bytes / steps, "kernels": out},
          open(sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/pmc_traffic.json", "w"), indent=1)
