"""MFMA-pipe utilisation per kernel from a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE and
SQ_BUSY_CYCLES.  usage: python tools_dev/pmc_mfma.py <db> [kernel_substring:K]
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD): busy cycles summed over the chip's SIMDs
against the cycles the GPU was active during the dispatch (so the DVFS clock cancels out)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
ev = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
t_min = 0
if len(sys.argv) > 2:
    kname, kth = sys.argv[2].rsplit(":", 1)
    ends = [r[0] for r in cur.execute("select d.end from %s d join %s s on d.kernel_id=s.id where s.%s like ? order by d.start" % (disp, sym, namecol), ("%" + kname + "%",))]
    t_min = ends[int(kth) - 1]
q = ("select s.%s, i.name, count(distinct d.id), sum(e.value), sum(d.end-d.start) * count(distinct d.id) / count(*) from %s d join %s s on d.kernel_id=s.id join %s e on e.event_id=d.event_id "
     "join %s i on i.id=e.pmc_id where d.start >= %d group by s.%s, i.name" % (namecol, disp, sym, ev, info, t_min, namecol))
agg = {}
for name, cname, n, v, dur in cur.execute(q):
    a = agg.setdefault(re.sub(r"\(.*", "", name), {})
    a[cname] = v; a["n"] = n; a["dur"] = dur
rows = []
for name, a in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in a or not a.get("GRBM_GUI_ACTIVE"):
        continue
    rows.append((a["dur"], name, a["n"], a["SQ_VALU_MFMA_BUSY_CYCLES"], a["GRBM_GUI_ACTIVE"], a.get("SQ_BUSY_CYCLES", 0)))
rows.sort(reverse=True)
# shader clock during a dispatch = GRBM_GUI_ACTIVE cycles (per XCD) / its duration -- but the counter window is wider than the
# dispatch by a fixed amount (the 5-us finalize kernels would run at "5.9 GHz"): that window is estimated from the shortest
# kernels, assumed to run at the nominal 2.4 GHz, and subtracted
short = [(gui / n / 8.0) - 2.4e3 * (dur / n / 1e3) for dur, name, n, mf, gui, sqb in rows if dur / n / 1e3 < 8.0]
window = sorted(short)[len(short) // 2] if short else 0.0
print("# counter window beyond a dispatch (median over the kernels under 8 us, at 2.4 GHz): %.0f cycles" % window)
print("%-62s %6s %10s %12s %10s %9s %9s" % ("kernel", "calls", "avg us", "MFMA busy", "GUI active", "MFMA util", "sclk GHz"))
out = {}
for dur, name, n, mf, gui, sqb in rows[:60]:
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
    util = mf / (1024.0 * gui / 8.0)
    # shader clock during the dispatch: GRBM_GUI_ACTIVE cycles (per XCD) / the dispatch's duration from the same row
    ghz = (gui / 8.0 - window * n) / dur
    print("%-62s %6d %10.1f %12.3e %10.3e %8.1f%% %9.3f" % (name[:62], n, dur / n / 1e3, mf / n, gui / n / 8.0, 100 * util, ghz))
    out[name] = {"launches": n, "avg_us": dur / n / 1e3, "mfma_busy": util, "sclk_ghz": round(ghz, 3)}
if len(sys.argv) > 3:
    import json
    import hashlib, os
    tp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scouter_amd", "tuning", "gfx950.json")
    sha = hashlib.sha256(open(tp, "rb").read()).hexdigest()[:16]
    json.dump({"command": "SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                          "SQ_BUSY_CYCLES -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-prof",
               "mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD)",
               "sclk_ghz": "GRBM_GUI_ACTIVE per XCD / dispatch duration (same rocprofv3 row)", "tuning_sha16": sha, "kernels": out},
              open(sys.argv[3], "w"), indent=1)
