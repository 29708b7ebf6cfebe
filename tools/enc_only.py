#!/usr/bin/env python3
"""Encoder alone (three conv blocks + BiLSTM): enc_only.py B T -> ms per call (HIP events, best of 5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import synthetic  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
model, ap, sd = bench.build_model(0, 1)
ids = torch.as_tensor(synthetic.t2_text_ids(5, B, T)).cuda()
model.encode(ids)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    model.encode(ids)
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
print("encoder B=%d T=%d: %.3f ms" % (B, T, best))
