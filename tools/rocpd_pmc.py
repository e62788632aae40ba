"""Per-kernel HBM traffic from two rocprofv3 PMC passes (rocpd sqlite): FETCH_SIZE and WRITE_SIZE.

usage: python tools/rocpd_pmc.py <fetch.db> <write.db> > profiles/rNN_pmc.md

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are in KB per dispatch; on gfx950
FETCH_SIZE counts the 128-byte requests of wide coalesced reads at 64 B, so it is DOUBLED here;
WRITE_SIZE is reported as is (uncalibrated in the guide).  Infinity-Cache hits are counted, not
excluded, so `traffic` is the L2<->fabric volume: an upper bound of the DRAM bytes.
"""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_stats import short


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(counter_value), sum(duration) from pmc_events where counter_name=? group by name",
                      (counter,)).fetchall()
    out = {}
    for name, n, tot, dur in rows:
        k = short(name)
        a = out.setdefault(k, [0, 0.0, 0.0])
        a[0] += n
        a[1] += tot
        a[2] += dur
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    if len(sys.argv) > 3:   # json: bytes per launch per kernel TEMPLATE (instances merged), read by bench.py
        import json
        import re
        agg = {}
        for k, (n, tot, _) in f.items():
            base = re.sub(r"<.*", "", k)
            a = agg.setdefault(base, [0, 0.0, 0.0])
            a[0] += n
            a[1] += tot * 2 * 1024.0
        for k, (n, tot, _) in w.items():
            base = re.sub(r"<.*", "", k)
            if base in agg:
                agg[base][2] += tot * 1024.0
        json.dump({k: {"launches": n, "read_bytes_per_launch": r / n, "write_bytes_per_launch": wv / n,
                       "bytes_per_launch": (r + wv) / n} for k, (n, r, wv) in agg.items()}, open(sys.argv[3], "w"), indent=1)
    print("| kernel | dispatches | read MB/launch (FETCH_SIZE x2) | write MB/launch (WRITE_SIZE) | total MB/launch | avg us (profiled) | TB/s |")
    print("|---|---|---|---|---|---|---|")
    keys = sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, [0, 0, 0])[1]))
    for k in keys[:40]:
        n, tot, dur = f[k]
        rd = tot * 2 / n / 1024.0
        wn, wtot, _ = w.get(k, [1, 0.0, 0.0])
        wr = wtot / max(wn, 1) / 1024.0
        us = dur / n / 1e3
        print(f"| `{k}` | {n} | {rd:.1f} | {wr:.1f} | {rd + wr:.1f} | {us:.1f} | {(rd + wr) * 1e6 / (us * 1e-6) / 1e12:.2f} |")


if __name__ == "__main__":
    main()
