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
    # steady state only: bench.py (JG_TRACE_MARK=1) brackets its timed steps with at::cuda's spin_kernel; without markers the whole trace
    # (model construction, warm-up) is summarised and `steps` must count every step in it
    marks = [r[0] for r in db.execute("select start from kernels where name like '%spin_kernel%' order by start").fetchall()]
    where, note = "", "whole trace"
    if len(marks) >= 2:
        where = f" where start > {marks[0]} and start < {marks[-1]} and name not like '%spin_kernel%'"
        note = f"steady-state window between the two marker kernels ({(marks[-1] - marks[0]) / 1e6:.2f} ms of wall time = {(marks[-1] - marks[0]) / 1e6 / steps:.2f} ms/step)"
    if "--by-queue" in sys.argv:        # one table per (queue, stream): which kernels a forked branch / side stream carries
        qs = db.execute("select queue_id, stream_id, count(*), sum(duration) from kernels" + where + " group by queue_id, stream_id order by sum(duration) desc").fetchall()
        for q, st, n, tot in qs:
            print(f"\n### queue {q} stream {st}: {n / steps:.0f} launches, {tot / 1e6 / steps:.3f} ms of kernel time per step\n")
            print("| kernel | calls/step | ms/step | avg us |")
            print("|---|---|---|---|")
            cond = (where + " and " if where else " where ") + f"queue_id = {q} and stream_id = {st}"
            for name, k, t, avg in db.execute("select name, count(*), sum(duration), avg(duration) from kernels" + cond + " group by name order by sum(duration) desc limit 40").fetchall():
                print(f"| `{short(name)}` | {k / steps:.1f} | {t / 1e6 / steps:.3f} | {avg / 1e3:.1f} |")
        return
    if "--by-grid" in sys.argv:         # one row per (kernel, launch geometry): under-filled grids and long launches of small problems stand out
        print("| kernel | queue | workgroups | threads | LDS KB | VGPRs | calls/step | avg us | ms/step |")
        print("|---|---|---|---|---|---|---|---|---|")
        q = ("select name, queue_id, (grid_x / workgroup_x) * (grid_y / workgroup_y) * (grid_z / workgroup_z), workgroup_x * workgroup_y * workgroup_z, lds_size, "
             "vgpr_count + accum_vgpr_count, count(*), avg(duration), sum(duration) from kernels" + where + " group by 1, 2, 3, 4, 5, 6 order by sum(duration) desc limit 120")
        for name, qu, wgs, th, lds, vg, k, avg, tot in db.execute(q).fetchall():
            print(f"| `{short(name)}` | {qu} | {wgs} | {th} | {lds / 1024:.0f} | {vg} | {k / steps:.2f} | {avg / 1e3:.1f} | {tot / 1e6 / steps:.3f} |")
        return
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels" + where +
                      " group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    nl = sum(r[1] for r in rows)
    print(f"total kernel time {total / 1e6:.2f} ms over {steps} step(s) = {total / 1e6 / steps:.2f} ms/step; {nl} launches = {nl / steps:.0f} per step; {note}\n")
    print("| kernel | calls | total ms | ms/step | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.2f} | {tot / 1e6 / steps:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.2f} |")


if __name__ == "__main__":
    main()
