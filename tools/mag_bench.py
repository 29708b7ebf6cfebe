#!/usr/bin/env python3
"""Timing of the mel -> linear magnitude kernel alone (B x 800 frames): mag_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

model, ap, sd = bench.build_model(0, 1)
for B in (64, 1):
    spec = torch.rand(B, 800, 80, device="cuda") * 8 - 4
    ap.spec_to_mag(spec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ap.spec_to_mag(spec)
    torch.cuda.synchronize()
    print("B=%d spec_to_mag %.1f us" % (B, (time.perf_counter() - t0) / 20 * 1e6))
