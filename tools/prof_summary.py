#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace sqlite database: per-kernel (name, grid) count / avg / min / total."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
q = ("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, "
     "sum(end-start)/1000.0 from kernels group by name, grid_x, grid_y order by 8 desc limit %d" % int(sys.argv[2] if len(sys.argv) > 2 else 40))
print("%-70s %9s %5s %5s %7s %9s %9s %11s" % ("kernel", "grid_x", "gy", "wg", "calls", "avg_us", "min_us", "total_us"))
for r in c.execute(q):
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-70s %9d %5d %5d %7d %9.2f %9.2f %11.1f" % (name[:70], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
