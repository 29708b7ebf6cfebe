#!/usr/bin/env python3
"""Timing of the postnet alone at B x F frames, for several F (tile quantisation of the conv launches):
postnet_bench.py [B] [F ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Fs = [int(a) for a in sys.argv[2:]] or [768, 800, 832, 1024]
model, ap, sd = bench.build_model(0, 1)
for F in Fs:
    mel = torch.rand(B, F, 80, device="cuda") * 8 - 4
    model.postnet(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.postnet(mel)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("B=%d F=%d: postnet %.3f ms, %.2f ns per frame, %d column tiles of 64" % (B, F, dt * 1e3, dt / (B * F) * 1e9, B * F // 64))
