#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the summaries kept under
profiles/: profile_collect.py <tag> [decoder steps profiled, default 384]

  <tag>_bench_kernel_stats.csv     kernel-trace statistics of the default bench command
  <tag>_bench_under_rocprof.json   the bench line printed under the profiler
  <tag>_decoder_pmc.csv / .json    HBM bytes per launch of the decoder step (64 x 128): (2 * FETCH_SIZE + WRITE_SIZE) KB
                                   (gfx950 counts a wide coalesced read at half its bytes), and per step
  <tag>_decoder_sq.csv             SQ counters per launch of the decoder step (MFMA busy cycles, wave cycles, ...)
  <tag>_gl_pmc.csv                 Griffin-Lim (64 x 800 frames): bytes and SQ counters per gl_fused_kernel launch
  <tag>_t1_batch1_kernel_stats.csv kernel-trace statistics of the Tacotron (v1) batch-1 run
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dec_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 384
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.environ.get("PROFILE_DST", os.path.join(ROOT, "profiles"))
os.makedirs(dst, exist_ok=True)


STEP_KERNELS = ("dual_kernel", "triple_kernel")  # the launches of a decoder step


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0] if name.startswith(STEP_KERNELS) or name.startswith("gl_fused") else name[:100]


def counters(subdir):
    """{kernel: {counter: [dispatches, total]}} of one --pmc pass."""
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(os.path.join(src, subdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def copy_stats(subdir, out_name):
    hits = glob.glob(os.path.join(src, subdir, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, out_name))
        print("wrote", out_name)


copy_stats("bench", tag + "_bench_kernel_stats.csv")
copy_stats("t1", tag + "_t1_batch1_kernel_stats.csv")
log = os.path.join(src, "bench_under_rocprof.log")
if os.path.exists(log):
    lines = [ln for ln in open(log) if ln.startswith('{"metric"')]
    if lines:
        open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(lines[-1])

# ---- decoder: HBM traffic per launch and per step
fetch, write = counters("dec_FETCH_SIZE"), counters("dec_WRITE_SIZE")
rows, total = [], 0.0
for k in sorted(fetch, key=lambda k: -fetch[k]["FETCH_SIZE"][1]):
    if not k.startswith(STEP_KERNELS):
        continue
    n, f = fetch[k]["FETCH_SIZE"]
    w = write.get(k, {}).get("WRITE_SIZE", [n, 0.0])[1]
    per = (2 * f + w) * 1024 / n
    total += (2 * f + w) * 1024
    rows.append((k, n, f / n, w / n, per))
if rows:
    with open(os.path.join(dst, tag + "_decoder_pmc.csv"), "w") as f:
        f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch,WRITE_SIZE_KB_per_dispatch,hbm_bytes_per_dispatch(2*FETCH+WRITE)*1024\n")
        for r in rows:
            f.write('"%s",%d,%.1f,%.1f,%.0f\n' % r)
    json.dump({"B": 64, "T": 128, "hbm_bytes_per_step": total / dec_steps, "steps_profiled": dec_steps,
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/dec_only.py 64 128 64; "
                      "per launch (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes (gfx950: FETCH_SIZE counts half of a wide "
                      "coalesced read), summed over every decoder launch of the run and divided by its steps"},
              open(os.path.join(dst, tag + "_decoder_pmc.json"), "w"))
    print("decoder HBM bytes per step: %.1f MB" % (total / dec_steps / 1e6))

sq = counters("dec_SQ")
if sq:
    names = sorted({c for k in sq for c in sq[k]})
    with open(os.path.join(dst, tag + "_decoder_sq.csv"), "w") as f:
        f.write("kernel,dispatches," + ",".join(n + "_per_dispatch" for n in names) + "\n")
        for k in sorted(sq, key=lambda k: -sq[k].get("SQ_BUSY_CYCLES", [0, 0])[1]):
            if k.startswith(STEP_KERNELS):
                n = max(v[0] for v in sq[k].values())
                f.write('"%s",%d,' % (k, n) + ",".join("%.0f" % (sq[k][c][1] / max(sq[k][c][0], 1)) for c in names) + "\n")
    print("wrote decoder sq")

# ---- Griffin-Lim
gf, gw, gs = counters("gl_FETCH_SIZE"), counters("gl_WRITE_SIZE"), counters("gl_SQ")
keys = [k for k in gf if "gl_fused" in k or "deemph" in k]
if keys:
    names = sorted({c for k in gs for c in gs[k]})
    with open(os.path.join(dst, tag + "_gl_pmc.csv"), "w") as f:
        f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch,WRITE_SIZE_KB_per_dispatch,hbm_bytes_per_dispatch(2*FETCH+WRITE)*1024,"
                + ",".join(n + "_per_dispatch" for n in names) + "\n")
        for k in keys:
            n, fv = gf[k]["FETCH_SIZE"]
            wv = gw.get(k, {}).get("WRITE_SIZE", [n, 0.0])[1]
            f.write('"%s",%d,%.1f,%.1f,%.0f,' % (k, n, fv / n, wv / n, (2 * fv + wv) * 1024 / n)
                    + ",".join("%.0f" % (gs.get(k, {}).get(c, [1, 0])[1] / max(gs.get(k, {}).get(c, [1, 0])[0], 1)) for c in names) + "\n")
    print("wrote gl pmc")
