#!/usr/bin/env python3
"""Achievable HBM bandwidth on this GPU with stock ops (calibration for roofline fractions)."""
import time
import torch

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for mb in (32, 76, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.rand(n, device="cuda"); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); ts = t(lambda: a.sum()); tm = t(lambda: a.mul_(1.0001))
    print("%5d MB  copy %.0f GB/s (r+w)  sum %.0f GB/s (r)  inplace-mul %.0f GB/s (r+w)" %
          (mb, 2 * n * 4 / tc / 1e9, n * 4 / ts / 1e9, 2 * n * 4 / tm / 1e9))
