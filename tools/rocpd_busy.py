"""Per-kernel MFMA-busy fraction from a rocprofv3 PMC pass (rocpd sqlite) with SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE (+ SQ_BUSY_CYCLES).
usage: python tools/rocpd_busy.py <busy.db>

Units (MI355X_MICROARCH.md, "Latency / throughput" table): SQ_VALU_MFMA_BUSY_CYCLES adds the issue cycles of every MFMA on every SIMD
(16 per v_mfma_f32_16x16x32 -- checked against the algorithmic MFMA count of the 512 -> 512 layer: 9.44e6 MFMAs x 16 = 1.51e8 =
the counter); the pmc_events table holds one row per hardware instance, so a dispatch's value is the SUM of its rows.  GRBM_GUI_ACTIVE is
reported per XCD (8): wall cycles of a dispatch = sum of its rows / 8, i.e. the effective clock = that / duration.
  MFMA busy            = MFMA busy cycles / (1024 SIMDs x wall cycles)            -- fraction of the MFMA issue slots of the launch
  MFMA busy vs nominal = MFMA busy cycles / (1024 SIMDs x duration x 2.4 GHz)     -- the same against the clock the 2.5 PFLOP/s peak assumes"""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_stats import short

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
durs = {short(n): (c, d) for n, c, d in db.execute("select name, count(*), sum(duration) from kernels group by name")}
agg = {}
for name, cn, nd, tot in rows:
    agg.setdefault(short(name), {})[cn] = (nd, tot)
print("| kernel | dispatches | avg us | MFMA busy cycles / dispatch | effective clock GHz | MFMA busy | MFMA busy vs 2.4 GHz nominal |")
print("|---|---|---|---|---|---|---|")
out = []
for k, a in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in a or "GRBM_GUI_ACTIVE" not in a or k not in durs:
        continue
    nd, mf = a["SQ_VALU_MFMA_BUSY_CYCLES"]
    _, gui = a["GRBM_GUI_ACTIVE"]
    nk, dur = durs[k]
    if mf <= 0 or nd == 0:
        continue
    us = dur / nk / 1e3
    wall = gui / 8.0 / nd
    out.append((dur, k, nd, us, mf / nd, wall / (us * 1e3), mf / nd / (1024.0 * wall), mf / nd / (1024.0 * us * 1e-6 * 2.4e9)))
for dur, k, nd, us, mf, clk, frac, fracn in sorted(out, reverse=True)[:25]:
    print(f"| `{k}` | {nd} | {us:.1f} | {mf:.3e} | {clk:.2f} | {frac:.3f} | {fracn:.3f} |")
