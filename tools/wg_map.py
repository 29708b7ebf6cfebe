#!/usr/bin/env python3
"""Where the workgroups of one decoder-step launch ran and for how long (tuning build: the library made with
-DTTSX_FA_TIMING as tts_amd/libttsx_dbg.so, run with TTSX_LIB=libttsx_dbg.so): wg_map.py B T grid n_first [pieces]
grid = workgroups of the launch to look at (e.g. 320 = fused attention + 256 staged pieces at B=64), n_first = how
many of them are the serial-path op.  Prints the lifetime of both kinds of workgroup and how they share CUs."""
import ctypes
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import _lib, synthetic  # noqa: E402

B, T, grid, n_first = (int(x) for x in sys.argv[1:5])
model, ap, sd = bench.build_model(0, 1)
if len(sys.argv) > 5:
    model.set_pieces(int(sys.argv[5]))
lib = _lib.load()
lib.ttsx_debug_wg_marks.restype = ctypes.c_int
lib.ttsx_debug_wg_marks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
ids = torch.as_tensor(synthetic.t2_text_ids(5, B, T)).cuda()
enc = model.encode(ids)
model.decode(enc, None, fixed_steps=64)
torch.cuda.synchronize()
lib.ttsx_debug_wg_marks(grid, n_first, None)
model.decode(enc, None, fixed_steps=64)
torch.cuda.synchronize()
out = np.zeros((2048, 4), dtype=np.int64)
lib.ttsx_debug_wg_marks(0, 0, out.ctypes.data)
m = out[:grid]
t0 = m[:, 0].min()
start, end = (m[:, 0] - t0) / 100.0, (m[:, 1] - t0) / 100.0
hw, xcc = m[:, 2], m[:, 3] & 0xF
cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
print("launch of %d workgroups (%d serial-path + %d others): ends %.2f us after its first workgroup starts" %
      (grid, n_first, grid - n_first, end.max()))
for name, sl in (("serial-path", slice(0, n_first)), ("others", slice(n_first, grid))):
    if sl.stop > sl.start:
        life = end[sl] - start[sl]
        print("  %-11s start %.2f..%.2f us, lifetime min %.2f / median %.2f / max %.2f us, last end %.2f" %
              (name, start[sl].min(), start[sl].max(), life.min(), np.median(life), life.max(), end[sl].max()))
per = defaultdict(list)
for i in range(grid):
    per[int(cu[i])].append(i)
kinds = defaultdict(int)
for c, ws in per.items():
    kinds[(sum(1 for w in ws if w < n_first), sum(1 for w in ws if w >= n_first))] += 1
print("  CUs used: %d; (serial-path, other) workgroups per CU -> number of CUs: %s" % (len(per), dict(sorted(kinds.items()))))
for k in sorted(kinds):
    ws = [w for c, v in per.items() if (sum(1 for w in v if w < n_first), sum(1 for w in v if w >= n_first)) == k for w in v]
    a = [end[w] - start[w] for w in ws if w < n_first]
    b = [end[w] - start[w] for w in ws if w >= n_first]
    print("    CUs with %s: serial-path lifetime %s, others %s" %
          (k, "%.2f" % np.mean(a) if a else "-", "%.2f (last end %.2f)" % (np.mean(b), max(end[w] for w in ws if w >= n_first)) if b else "-"))
