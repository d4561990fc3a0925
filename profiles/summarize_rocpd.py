#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the per-kernel summary
table we keep under profiles/.   usage: summarize_rocpd.py results.db > summary.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# rocprofv3 --kernel-trace --stats : per-kernel summary (durations in us)")
    print("%-100s %6s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        print("%-100s %6d %12.1f %10.1f %10.1f %10.1f %6.2f%%" % (name[:100], n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    print()
    print("# launch geometry / resources of the first dispatch of each kernel")
    for r in cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size "
                         "from kernels group by name"):
        print("%-100s grid=(%d,%d,%d) wg=%d lds=%d vgpr=%d sgpr=%d scratch=%d" % ((r[0][:100],) + r[1:]))


if __name__ == "__main__":
    main(sys.argv[1])
