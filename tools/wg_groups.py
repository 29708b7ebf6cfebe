#!/usr/bin/env python3
"""Start / end of the workgroups of one decoder-step launch by block-index range (tuning build, as tools/wg_map.py):
wg_groups.py B T grid n_first bound [bound ...]   e.g. 64 128 512 64 64 256: attention | staged pieces | late LSTM"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import _lib, synthetic  # noqa: E402

B, T, grid, n_first = (int(x) for x in sys.argv[1:5])
bounds = [0] + [int(x) for x in sys.argv[5:]] + [grid]
model, ap, sd = bench.build_model(0, 1)
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
print("launch of %d workgroups ends %.2f us after its first workgroup starts" % (grid, end.max()))
for lo, hi in zip(bounds[:-1], bounds[1:]):
    s, e = start[lo:hi], end[lo:hi]
    print("  blocks %4d..%4d: start %.2f / %.2f / %.2f us (min / median / max), end %.2f / %.2f / %.2f, lifetime median %.2f" %
          (lo, hi - 1, s.min(), np.median(s), s.max(), e.min(), np.median(e), e.max(), np.median(e - s)))
