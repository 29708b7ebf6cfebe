#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel from its CSV output:
pmc_traffic.py <dir> <counter>  ->  kernel, dispatches, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-90s %7d %14.1f" % (k[:90], n, v / n))
