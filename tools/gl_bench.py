#!/usr/bin/env python3
"""Griffin-Lim only micro-benchmark (B x F frames, 60 iterations)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd.audio import AudioProcessor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 800
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ap = AudioProcessor(sample_rate=22050, num_mels=80, min_level_db=-100, frame_shift_ms=12.5, frame_length_ms=50,
                    ref_level_db=20, num_freq=1025, power=1.5, preemphasis=0.98, signal_norm=True,
                    symmetric_norm=True, max_norm=4, mel_fmin=0.0, mel_fmax=8000.0, griffin_lim_iters=60,
                    verbose=False)
mag = torch.rand(B, F, 1025, device="cuda") * 3
ph = torch.rand(B, F, 1025, device="cuda")
ap.griffin_lim_batch(mag, ph)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ap.griffin_lim_batch(mag, ph)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
frames = B * F
print("GL B=%d F=%d: %.2f ms  -> %.2f ns/frame-iteration, %.1f GB/s algorithmic" %
      (B, F, dt * 1e3, dt / frames / 61 * 1e9, frames * 383200 / dt / 1e9))
