#!/usr/bin/env python3
"""Stage timings of the Tacotron (v1) path on one GPU: encoder CBHG, greedy decoder (per step),
post-CBHG + last_linear, Griffin-Lim with the v1 STFT parameters."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_amd import synthetic  # noqa: E402
from tts_amd.audio import AudioProcessor  # noqa: E402
from tts_amd.tacotron import Tacotron  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    sd = synthetic.t1_state_dict(seed=0, gains=dict(stop=0.0, stop_bias=-20.0, v=2.0))
    m = Tacotron(61, 256, 1025, 80, 2, None, 5)
    m.load_state_dict(sd)
    m.cuda().eval()
    m.decoder.max_decoder_steps = steps
    ids = torch.as_tensor(np.concatenate([synthetic.t1_text_ids(b, T) for b in range(B)])).cuda()
    t_enc, enc = timed(lambda: m.encode(ids))
    t_dec, dec = timed(lambda: m.decode(enc), 3)
    mel, n = dec[0], dec[3]
    t_post, lin = timed(lambda: m.postnet(mel))
    ap = AudioProcessor(sample_rate=22050, num_mels=80, min_level_db=-100, frame_shift_ms=12.5, frame_length_ms=50,
                        ref_level_db=20, num_freq=1025, power=1.5, preemphasis=0.98, signal_norm=True,
                        symmetric_norm=False, max_norm=1, clip_norm=True, griffin_lim_iters=60, tacotron_v1=True,
                        verbose=False)
    ph = torch.rand(B, lin.shape[1], 1025, device="cuda")
    t_gl, _ = timed(lambda: ap.griffin_lim_batch(ap.spec_to_mag(lin, mel=False), ph), 3)
    s = int(n.max())
    frames = B * lin.shape[1]
    print(json.dumps(dict(B=B, T=T, steps=s, frames=frames, encoder_ms=t_enc * 1e3, decoder_ms=t_dec * 1e3,
                          decoder_us_per_step=t_dec / s * 1e6, postnet_ms=t_post * 1e3, griffin_lim_ms=t_gl * 1e3,
                          frames_per_s=frames / (t_enc + t_dec + t_post + t_gl))))


if __name__ == "__main__":
    main()
