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


def note_name(k):
    """rocprof instance name (template arguments) -> the name the dispatch reports through jg_last_kernel() (bench.py's roofline rows)"""
    import re
    m = re.match(r"([a-z_0-9]+)<(?:bf16|f16)((?:,[^,>]+)*)>", k)
    if not m:
        return None
    base, a = m.group(1), m.group(2).split(",")[1:]
    if base == "wgrad3x3_halo_kernel" and len(a) >= 3:
        return {("16", "2", "2"): "wgrad3x3_halo_kernel<16 rows,64 co>", ("8", "4", "2"): "wgrad3x3_halo_kernel<8 rows,128 co>",
                ("8", "4", "1"): "wgrad3x3_halo_kernel<8 rows,64 co,4 waves>"}.get(tuple(a[:3]))
    if base == "conv3x3_halo_kernel" and len(a) >= 2:
        if len(a) >= 7 and "Lb1E" in a[6]:            # PHASE = true (sub-pixel form)
            return "conv3x3_halo_kernel<subpixel>"
        return {("256", "512"): "conv3x3_halo_kernel<256-wide,8 waves>", ("128", "512"): "conv3x3_halo_kernel<128-wide,8 waves>",
                ("128", "256"): "conv3x3_halo_kernel<128-wide,4 waves>", ("64", "256"): "conv3x3_halo_kernel<64-wide>"}.get(tuple(a[:2]))
    if base == "conv_nt_glds_kernel" and len(a) >= 5:
        return "conv_nt_glds_kernel<" + ",".join(a[:5]) + ">"
    if base == "wgrad_kxk_halo_kernel":
        return "wgrad_kxk_halo_kernel<7x7,2 tap rows>"
    if base == "wgrad_tn_tr_kernel" and a:
        return "wgrad_tn_tr_kernel<" + a[0] + ">"
    return None


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
        # the same per dispatch-reported instance (several template instances -- PIPE on / off, pad modes -- can share a name)
        for k, (n, tot, _) in f.items():
            nm = note_name(k)
            if nm:
                a = agg.setdefault(nm, [0, 0.0, 0.0])
                a[0] += n
                a[1] += tot * 2 * 1024.0
                a[2] += w.get(k, [0, 0.0, 0.0])[1] * 1024.0
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
