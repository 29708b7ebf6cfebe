#!/usr/bin/env python3
"""Phase timestamps of the fused attention kernel (tuning build: make the library with -DTTSX_FA_TIMING
as tts_amd/libttsx_dbg.so and run with TTSX_LIB=libttsx_dbg.so): fa_timing.py B T [pieces [split]]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import _lib, synthetic  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
model, ap, sd = bench.build_model(0, 1)
if len(sys.argv) > 3:
    model.set_pieces(int(sys.argv[3]))
if len(sys.argv) > 4:
    model.set_plan(-1, int(sys.argv[4]), -1)
ids = torch.as_tensor(synthetic.t2_text_ids(5, B, T)).cuda()
enc = model.encode(ids)
model.decode(enc, None, fixed_steps=64)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 16)()
lib = _lib.load()
lib.ttsx_debug_fa_timing.restype = ctypes.c_int
lib.ttsx_debug_fa_timing(out)
t = list(out)
names = ["loads+stage", "q sum", "energies (MFMA)", "normalise", "context FMAs", "reduce+store"]
print("B=%d T=%d" % (B, T), ", ".join("%s %.2f us" % (n, (t[i + 1] - t[i]) / 100.0) for i, n in enumerate(names)),
      "| total %.2f us" % ((t[6] - t[0]) / 100.0))
g = t[8:12]
if g[3] > g[0] > 0:
    print("decoder-LSTM GEMM workgroup 0: K loop %.2f us, reduce barrier %.2f us, epilogue %.2f us | total %.2f us"
          % ((g[1] - g[0]) / 100.0, (g[2] - g[1]) / 100.0, (g[3] - g[2]) / 100.0, (g[3] - g[0]) / 100.0))
