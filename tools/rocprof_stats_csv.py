#!/usr/bin/env python3
"""Export the per-kernel summary of a `rocprofv3 --kernel-trace --stats` run (rocpd SQLite output) as CSV:
   python tools/rocprof_stats_csv.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
Durations are nanoseconds, as in rocprofv3's own kernel_stats.csv."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, calls, tot, avg, mn, mx in rows:
        print('"%s",%d,%d,%.3f,%.2f,%d,%d' % (name, calls, tot, avg, 100.0 * tot / total, mn, mx))


if __name__ == "__main__":
    main()
