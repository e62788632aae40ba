"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd database (dev tool).  usage: python tools/rocpd_counters.py <file.db> [name filter]"""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_stats import short

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
durs = {short(n): (c, d) for n, c, d in db.execute("select name, count(*), sum(duration) from kernels group by name")}
agg = {}
for name, cn, nd, tot in rows:
    agg.setdefault(short(name), {})[cn] = tot / max(nd, 1)
for k, a in sorted(agg.items(), key=lambda kv: -durs.get(kv[0], (0, 0))[1]):
    if flt and flt not in k:
        continue
    nk, dur = durs.get(k, (1, 0))
    print(f"{k}: dispatches {nk}, avg {dur / nk / 1e3:.1f} us")
    for cn, v in sorted(a.items()):
        print(f"    {cn:32s} {v:.4e}")
