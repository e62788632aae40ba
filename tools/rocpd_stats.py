"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a markdown table.

usage: python tools/rocpd_stats.py gpurun_out/x_prof/x_results.db [steps_in_trace] > profiles/rNN_x.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*?)EEv", name)
    if m:
        targs = m.group(2).replace("DF16b", "bf16").replace("DF16_", "f16")
        targs = re.sub(r"Li(\d+)E", r",\1", targs)
        return f"{m.group(1)}<{targs}>"
    name = re.sub(r"<.*", "<...>", name) if len(name) > 90 else name
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"total kernel time {total / 1e6:.2f} ms over {steps} step(s) = {total / 1e6 / steps:.2f} ms/step\n")
    print("| kernel | calls | total ms | ms/step | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.2f} | {tot / 1e6 / steps:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.2f} |")


if __name__ == "__main__":
    main()
