"""How much of a rocprofv3 kernel trace (rocpd sqlite) runs with two or more kernels in flight (multi-stream overlap).
usage: python tools/rocpd_overlap.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
rows = db.execute("select start, end, name, queue_id, stream_id from kernels order by start").fetchall() if "stream_id" in cols else \
    db.execute("select start, end, name, queue_id, 0 from kernels order by start").fetchall()
ev = []
for s, e, n, q, st in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
active, last, busy1, busy2 = 0, None, 0, 0
for t, d in ev:
    if last is not None and active >= 1:
        busy1 += t - last
        if active >= 2:
            busy2 += t - last
    active += d
    last = t
print(f"kernels {len(rows)}  time with >=1 kernel {busy1 / 1e6:.2f} ms, with >=2 kernels {busy2 / 1e6:.2f} ms, span {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms")
qs = {}
for s, e, n, q, st in rows:
    qs.setdefault((q, st), [0, 0])
    qs[(q, st)][0] += 1
    qs[(q, st)][1] += e - s
for k, v in qs.items():
    print("queue/stream", k, "kernels", v[0], "sum ms", v[1] / 1e6)
