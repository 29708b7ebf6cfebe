#!/usr/bin/env python3
"""Summary of the counter passes of tools/profile_team.sh: team_pmc_collect.py <tag>
-> <tag>_team_pmc.csv: one row per (case, team kernel): dispatches, steps per dispatch, kernel time per dispatch,
HBM bytes per dispatch and per step ((2 * FETCH_SIZE + WRITE_SIZE) KB: gfx950 counts a wide coalesced read at half
its bytes), SQ counters per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_%s_team" % tag)
dst = os.environ.get("PROFILE_DST", os.path.join(ROOT, "profiles"))
os.makedirs(dst, exist_ok=True)
CASES = (("t2_100_800", 800), ("t2_1000_2000", 2000), ("t1_100_200", 200))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counters(subdir):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(os.path.join(src, subdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def durations(subdir):
    out = {}
    for path in glob.glob(os.path.join(src, subdir, "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                out[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]))
    return out


rows, sq_names = [], []
for case, steps in CASES:
    fetch, write, sq, kt = (counters(case + "_FETCH_SIZE"), counters(case + "_WRITE_SIZE"), counters(case + "_SQ"),
                            durations(case + "_kt"))
    for k in fetch:
        if "team_kernel" not in k:
            continue
        n, f = fetch[k]["FETCH_SIZE"]
        w = write.get(k, {}).get("WRITE_SIZE", [n, 0.0])[1]
        per = (2 * f + w) * 1024 / n
        s = {c: v[1] / max(v[0], 1) for c, v in sq.get(k, {}).items()}
        sq_names = sorted(set(sq_names) | set(s))
        rows.append((case, k, n, steps, kt.get(k, (0, 0.0))[1] / 1e3, f / n, w / n, per, per / steps, s))
with open(os.path.join(dst, tag + "_team_pmc.csv"), "w") as f:
    f.write("case,kernel,dispatches,steps_per_dispatch,kernel_us_per_dispatch,FETCH_SIZE_KB_per_dispatch,"
            "WRITE_SIZE_KB_per_dispatch,hbm_bytes_per_dispatch,hbm_bytes_per_step,"
            + ",".join(n + "_per_dispatch" for n in sq_names) + "\n")
    for r in rows:
        f.write('%s,"%s",%d,%d,%.1f,%.1f,%.1f,%.0f,%.0f,' % r[:9] + ",".join("%.0f" % r[9].get(n, 0) for n in sq_names) + "\n")
print(open(os.path.join(dst, tag + "_team_pmc.csv")).read())
