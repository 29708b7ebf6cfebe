#!/usr/bin/env python3
"""Does splitting a batch over concurrent streams pay?  dec_dual.py B T steps NSPLIT
Runs the decoder loop on B utterances as NSPLIT sub-batches, each on its own decoder handle, HIP
stream and host thread, and prints the wall time per step next to the single-batch figure."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tts_amd import synthetic  # noqa: E402

B, T, S, N = (int(x) for x in sys.argv[1:5])
models = [bench.build_model(0, 1)[0] for _ in range(N)]
ids = torch.as_tensor(synthetic.t2_text_ids(5, B, T)).cuda()
enc = models[0].encode(ids)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(N)]
parts = [enc[i * (B // N):(i + 1) * (B // N)].contiguous() for i in range(N)]


def worker(i, steps):
    with torch.cuda.stream(streams[i]):
        models[i].decode(parts[i], None, fixed_steps=steps)
        streams[i].synchronize()


def run_split(steps):
    th = [threading.Thread(target=worker, args=(i, steps)) for i in range(N)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def run_single(steps):
    t0 = time.perf_counter()
    models[0].decode(enc, None, fixed_steps=steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for fn, name in ((run_single, "single"), (run_split, "split x%d" % N)):
    fn(S)
    best = min(fn(S) for _ in range(3))
    print("B=%d T=%d %s: %.2f us/step all-in (%d steps)" % (B, T, name, best / S * 1e6, S), flush=True)
