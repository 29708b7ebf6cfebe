#!/usr/bin/env python3
"""Free-running batch-1 decode (late-stop checkpoint, stop near step 309) under different polling settings:
steps per graph / steps kept in flight -> decoder time, steps enqueued."""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import _lib, synthetic  # noqa: E402

model, ap, sd = bench.build_model(0, 1, gains=synthetic.T2_LATE_STOP_GAINS)
model.decoder.max_decoder_steps = 3000
ids = torch.as_tensor(synthetic.t2_text_ids(99, 1, 100)).cuda()
enc = model.encode(ids)
lib = _lib.load()
for chunk, ahead in ((4, 8), (8, 8), (8, 12), (8, 16), (16, 16), (16, 24), (32, 32), (32, 48), (32, 64)):
    _lib.check(lib.ttsx_t2_decoder_set_polling(model._dec_handle, chunk, ahead), "set_polling")
    with contextlib.redirect_stdout(io.StringIO()):
        model.decode(enc, None)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            mel, align, stop, n_steps, steps = model.decode(enc, None)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
    print("chunk %2d ahead %2d: %.3f ms, n_steps %d, enqueued %d (%.2f us per needed step)" %
          (chunk, ahead, best * 1e3, int(n_steps[0]), model.last_steps_launched(), best * 1e6 / int(n_steps[0])))
