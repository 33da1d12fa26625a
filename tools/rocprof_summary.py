#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db, or *_kernel_stats.csv) into the
small text table kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_bench_kernel_stats.md "command line"
"""
import sqlite3
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    db = sqlite3.connect(src)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    extra = {}
    for name, vg, sg, lds, scr, wg, gx in cur.execute(
            "select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(workgroup_x), max(grid_x) from kernels group by name"):
        extra[name] = (vg, sg, lds, scr, wg, gx)
    with open(dst, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n")
        if cmd:
            f.write("command: `%s`\n\n" % cmd)
        f.write("| kernel | calls | total ms | avg ms | % | VGPR | SGPR | LDS B | scratch B | block | grid |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for name, calls, total, avg, pct in rows:
            e = extra.get(name, ("",) * 6)
            f.write("| `%s` | %d | %.3f | %.4f | %.2f | %s | %s | %s | %s | %s | %s |\n" % (name.split("(")[0], calls, total / 1e6 if total > 1e7 else total / 1e3,
                                                                                 avg / 1e6 if total > 1e7 else avg / 1e3, pct, *e))
    print(open(dst).read())


if __name__ == "__main__":
    main()
