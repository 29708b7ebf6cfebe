#!/usr/bin/env python3
"""Timestamps inside one round of the Griffin-Lim iteration kernel (tuning build: griffin_lim.hip compiled with
-DTTSX_GL_TIMING and linked as tts_amd/libttsx_dbg.so; run with TTSX_LIB=libttsx_dbg.so): gl_timing.py [B] [F]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import _lib  # noqa: E402
from tts_amd.audio import AudioProcessor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 800
ap = AudioProcessor(sample_rate=22050, num_mels=80, min_level_db=-100, frame_shift_ms=12.5, frame_length_ms=50,
                    ref_level_db=20, num_freq=1025, power=1.5, preemphasis=0.98, signal_norm=True,
                    symmetric_norm=True, max_norm=4, mel_fmin=0.0, mel_fmax=8000.0, griffin_lim_iters=60,
                    verbose=False)
mag = torch.rand(B, F, 1025, device="cuda") * 3
ph = torch.rand(B, F, 1025, device="cuda")
for _ in range(2):
    ap.griffin_lim_batch(mag, ph)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 32)()
lib = _lib.load()
lib.ttsx_debug_gl_timing.restype = ctypes.c_int
assert lib.ttsx_debug_gl_timing(out) == 0
t = list(out)
names = ["analysis loads + window products", "forward transform", "bin pairs", "inverse transform",
         "window + frame to LDS", "barrier (slowest wave of the round)", "overlap-add, normalise, write"]
print("wave 0 of the middle workgroup, second round of the last iteration (us):")
for i, n in enumerate(names):
    print("  %-40s +%.2f  (at %.2f)" % (n, (t[i + 1] - t[i]) / 100.0, (t[i + 1] - t[0]) / 100.0))
