#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd / SQLite) result: per-kernel stats like `--stats`, plus a
per-(kernel, grid) breakdown.  Usage: tools/rocprof_summary.py results.db > profiles/xxx.txt"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# KERNEL_DISPATCH stats (durations in ns)")
    print("%-78s %8s %14s %12s %7s" % ("Name", "Calls", "TotalDur(ns)", "AvgDur(ns)", "Pct"))
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-78s %8d %14.0f %12.1f %7.2f" % (name[:78], calls, total * 1000, avg * 1000, pct))
    print("\n# per (kernel, grid size): calls, avg / min / max duration (ns), LDS bytes, VGPR+AGPR")
    q = ("select name, grid_x, workgroup_x, count(*), avg(duration), min(duration), max(duration), max(lds_size), "
         "max(vgpr_count), max(accum_vgpr_count) from kernels group by name, grid_x order by name, grid_x")
    for r in cur.execute(q):
        print("%-62s grid=%-8d wg=%-4d n=%-5d avg=%-9.0f min=%-8d max=%-8d lds=%-6d vgpr=%d+%d" % (
            (r[0][:62],) + tuple(r[1:])))


if __name__ == "__main__":
    main(sys.argv[1])
