#!/usr/bin/env python3
"""Timestamps inside the Tacotron (v1) team kernel at step 50 (tuning build: t1_team.hip compiled with
-DTTSX_TEAM_TIMING and linked as tts_amd/libttsx_dbg.so; run with TTSX_LIB=libttsx_dbg.so): team_timing.py T"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import _lib, synthetic  # noqa: E402
from tts_amd.tacotron import Tacotron  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sd = synthetic.t1_state_dict(seed=0, gains=dict(stop=0.0, stop_bias=-20.0, v=2.0))
m = Tacotron(61, 256, 1025, 80, 2, None, 5)
m.load_state_dict(sd)
m.cuda().eval()
m.decoder.max_decoder_steps = 100
ids = torch.as_tensor(synthetic.t1_text_ids(0, T)).cuda()
enc = m.encode(ids)
for _ in range(3):
    m.decode(enc)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 320)()
lib = _lib.load()
lib.ttsx_debug_team_timing.restype = ctypes.c_int
assert lib.ttsx_debug_team_timing(out) == 0
t = list(out)
row = ["frames arrive + stop rule", "prenet 1 published", "x1 arrives", "prenet 2", "attention GRU + query share",
       "h_att + alignment arrive", "GRU 1 published", "d1 arrives", "GRU 2 published", "d2 arrives", "frames published"]
print("row workgroup 0 (us since the top of step 50):")
for i, n in enumerate(row):
    print("  %-32s +%.2f  (at %.2f)" % (n, (t[i + 1] - t[i]) / 100.0, (t[i + 1] - t[0]) / 100.0))
att = [(17, "loop top"), (18, "query shares arrive"), (19, "energies"), (21, "sigmoids published"),
       (20, "all sigmoids seen, alignment state updated"), (16, "conv + location linear"), (22, "d2 + frames arrive")]
print("attention workgroup 0 (us relative to the row workgroup's step top):")
prev = t[17]
for i, n in att:
    print("  %-44s +%.2f  (at %.2f)" % (n, (t[i] - prev) / 100.0, (t[i] - t[0]) / 100.0))
    prev = t[i]
pub = [(t[64 + g] - t[0]) / 100.0 for g in range(32)]
got = [(t[128 + g] - t[0]) / 100.0 for g in range(32)]
bar = [(t[192 + g] - t[0]) / 100.0 for g in range(32)]
print("d1 published by workgroup g at:      " + " ".join("%.2f" % v for v in pub))
print("d1 + h1 seen by wave 0 of g at:      " + " ".join("%.2f" % v for v in got))
print("workgroup g past the barrier at:     " + " ".join("%.2f" % v for v in bar))
