#!/usr/bin/env python3
"""Run only the Tacotron2 decoder loop (for kernel profiles and step timing): dec_only.py B T steps [fused split share [team [fusion [pieces]]]]
(the optional launch-plan overrides of Tacotron2.set_plan / set_team: -1 by shape, 0 off, 1 on).
Prints the marginal time per step (difference between a run of `steps` and one of `steps`/2 steps,
so that per-run set-up cancels) and the all-in figure."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import synthetic  # noqa: E402

B, T, S = (int(x) for x in sys.argv[1:4])
plan = [int(x) for x in sys.argv[4:7]] + [-1] * (3 - len(sys.argv[4:7]))
model, ap, sd = bench.build_model(0, 1)
model.set_plan(*plan)
if len(sys.argv) > 7:
    model.set_team(int(sys.argv[7]))
if len(sys.argv) > 8:
    model.set_fusion(int(sys.argv[8]))
if len(sys.argv) > 9:
    model.set_pieces(int(sys.argv[9]))
ids = torch.as_tensor(synthetic.t2_text_ids(5, B, T)).cuda()
enc = model.encode(ids)


def run(steps, reps=3):
    model.decode(enc, None, fixed_steps=steps)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        model.decode(enc, None, fixed_steps=steps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


full, half = run(S), run(S // 2)
print("B=%d T=%d plan=%s: marginal %.2f us/step, all-in %.2f us/step (%d steps), fixed %.2f ms" %
      (B, T, plan, (full - half) / (S - S // 2) * 1e6, full / S * 1e6, S,
       (2 * half - full) * 1e3))
