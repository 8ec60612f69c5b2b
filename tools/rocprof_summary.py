#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel call count / total / average duration,
and, if the run collected PMC counters, the per-kernel average of each counter.

    python tools/rocprof_summary.py gpurun_out/prof_stats/r01_results.db [more.db ...] > profiles/r01_x.txt
"""
import sqlite3
import sys


def short(name, n=70):
    name = name.replace("void ", "")
    for cut in ("(", "<float, 4, at::native::templates"):
        pass
    if name.startswith("at::native") or name.startswith("__amd"):
        name = name.split("<")[0]
    else:
        name = name.split("(")[0]
    return name[:n]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print(f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
            for name, calls, total, avg, pct in rows[:12]:
                print(f"{short(name):72s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
        try:  # per launch shape: bench.py launches the same kernels in stream mode, block mode and the parity gate
            rows = list(cur.execute(
                "select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), min(duration), "
                "max(duration) from kernels where name like '%earhip%' group by name, grid_x, grid_y, grid_z, workgroup_x "
                "order by sum(duration) desc"))
        except sqlite3.OperationalError:
            rows = []
        if rows:
            print(f"{'kernel (per launch shape)':44s} {'grid':>22s} {'wg':>5s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
            for name, gx, gy, gz, wx, n, tot, avg, mn, mx in rows[:16]:
                print(f"{short(name, 44):44s} {str((gx, gy, gz)):>22s} {wx:5d} {n:6d} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f}")
        try:
            rows = list(cur.execute(
                "select kernel_name, counter_name, count(*), avg(value), avg(duration), "
                "max(vgpr_count), max(sgpr_count), max(lds_block_size), max(grid_size), max(workgroup_size) "
                "from counters_collection group by kernel_name, counter_name"))
        except sqlite3.OperationalError:
            rows = []
        if rows:
            print(f"{'kernel':40s} {'counter':24s} {'n':>4s} {'avg_value':>16s} {'avg_ns':>10s}  vgpr sgpr lds grid wg")
            for k, c, n, v, d, vg, sg, lds, grid, wg in rows:
                if not short(k).startswith("earhip"):
                    continue
                print(f"{short(k, 40):40s} {c:24s} {n:4d} {v:16.1f} {d:10.0f}  {vg} {sg} {lds} {grid} {wg}")
        db.close()


if __name__ == "__main__":
    main()
