#!/usr/bin/env python3
"""Encoder alone (three conv blocks + BiLSTM): enc_only.py B T [scan] -> ms per call (HIP events, best of 5);
scan: -1 by shape, 0 chain of step launches, 1 persistent scan."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import synthetic  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
model, ap, sd = bench.build_model(0, 1)
if len(sys.argv) > 3:
    model.set_encoder_plan(int(sys.argv[3]))
ids = torch.as_tensor(synthetic.t2_text_ids(5, B, T)).cuda()
ref = model.encode(ids)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = model.encode(ids)
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
model.set_encoder_plan(0)
chain = model.encode(ids)
print("encoder B=%d T=%d: %.3f ms; max |scan - chain| = %.2e" % (B, T, best, float((out - chain).abs().max())))
