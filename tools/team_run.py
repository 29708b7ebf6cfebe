#!/usr/bin/env python3
"""A fixed number of batch-1 decoder runs through the persistent team kernels, for counter passes:
team_run.py t2 T steps [runs]   (Tacotron2: t2_team_kernel<128,1> for T <= 128, <256,1> up to 256, <256,4> beyond)
team_run.py t1 T steps [runs]   (Tacotron v1: t1_team_kernel)
Every run is one launch of the team kernel, so a counter divided by the dispatch count is a per-run figure and
divided by `steps` again a per-step figure."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import synthetic  # noqa: E402

which, T, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 4
if which == "t2":
    import bench
    model, _, _ = bench.build_model(0, 1)
    enc = model.encode(torch.as_tensor(synthetic.t2_text_ids(5, 1, T)).cuda())

    def run():
        return model.decode(enc, None, fixed_steps=steps)
else:
    from tts_amd.tacotron import Tacotron
    sd = synthetic.t1_state_dict(seed=0, gains=dict(stop=0.0, stop_bias=-20.0, v=2.0))   # runs to max_decoder_steps
    model = Tacotron(61, 256, 1025, 80, 2, None, 5)
    model.load_state_dict(sd)
    model.cuda().eval()
    model.decoder.max_decoder_steps = steps
    enc = model.encode(torch.as_tensor(np.concatenate([synthetic.t1_text_ids(0, T)])).cuda())

    def run():
        with contextlib.redirect_stdout(io.StringIO()):
            return model.decode(enc)

run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(runs - 1):
    run()
torch.cuda.synchronize()
print("%s T=%d steps=%d: %.2f us/step all-in over %d runs" %
      (which, T, steps, (time.perf_counter() - t0) / max(runs - 1, 1) / steps * 1e6, runs - 1))
