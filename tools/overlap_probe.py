#!/usr/bin/env python3
"""Do the decoder loop (latency-bound launches) and Griffin-Lim (VALU-bound, fills the chip) overlap
when issued from two host threads on two HIP streams?  Prints each alone and both together."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import synthetic  # noqa: E402

B, T, S = 64, 128, 800
model, ap, sd = bench.build_model(0, 1)
ids = torch.as_tensor(synthetic.t2_text_ids(5, B, T)).cuda()
enc = model.encode(ids)
mel, *_ = model.decode(enc, None, fixed_steps=S)
post = model.postnet(mel)
mag = ap.spec_to_mag(post)
phase = torch.rand(B, S, 1025, device="cuda")
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def dec():
    with torch.cuda.stream(s1):
        model.decode(enc, None, fixed_steps=S)
        s1.synchronize()


def gl():
    with torch.cuda.stream(s2):
        ap.griffin_lim_batch(mag, phase)
        s2.synchronize()


def timed(fns):
    best = 1e9
    for _ in range(3):
        th = [threading.Thread(target=f) for f in fns]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


dec(), gl()
print("decoder alone %.2f ms, griffin-lim alone %.2f ms, both concurrently %.2f ms" % (timed([dec]), timed([gl]), timed([dec, gl])), flush=True)
