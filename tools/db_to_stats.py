#!/usr/bin/env python3
"""Kernel statistics of a rocprofv3 --kernel-trace sqlite database in the column layout of rocprofv3's own
kernel_stats.csv (Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev): db_to_stats.py <db> > out.csv"""
import csv
import math
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "avg((end-start)*1.0*(end-start)) from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows) or 1
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for name, n, tot, avg, mn, mx, sq in rows:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    w.writerow([name, n, tot, "%.3f" % avg, "%.4f" % (100.0 * tot / total), mn, mx, "%.3f" % math.sqrt(max(sq - avg * avg, 0.0))])
