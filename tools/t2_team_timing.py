#!/usr/bin/env python3
"""Timestamps inside the Tacotron2 team kernel at step 50 (tuning build: t2_team.hip compiled with
-DTTSX_TEAM_TIMING and linked as tts_amd/libttsx_dbg.so; run with TTSX_LIB=libttsx_dbg.so): t2_team_timing.py T"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import _lib, synthetic  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
model, ap, sd = bench.build_model(0, 1)
ids = torch.as_tensor(synthetic.t2_text_ids(5, 1, T)).cuda()
enc = model.encode(ids)
for _ in range(3):
    model.decode(enc, None, fixed_steps=100)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 64)()
lib = _lib.load()
lib.ttsx_debug_t2_team_timing.restype = ctypes.c_int
assert lib.ttsx_debug_t2_team_timing(out) == 0
t = list(out)
names = {0: "proj: h_d + ctx of step 49 arrived", 1: "proj: x1 published", 2: "proj: x1 arrived", 3: "proj: x2 published",
         8: "att: x2 arrived", 9: "att: h_a published", 10: "att: gate sums ready", 12: "att: after the barrier", 16: "energy: h_a arrived", 17: "energy: partial energies published",
         18: "energy: all partial energies arrived", 19: "energy: alignment state updated", 24: "ctx: partial energies arrived",
         25: "ctx: context published", 32: "dec: h_a arrived", 33: "dec: partial energies arrived", 34: "dec: h_d published"}
t0 = t[0]
for i in sorted(names, key=lambda i: t[i]):
    print("%8.2f us  %s" % ((t[i] - t0) / 100.0, names[i]))
