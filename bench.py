#!/usr/bin/env python3
"""Benchmark of the synthesis hot path: mel frames/s of Tacotron2 + Griffin-Lim.

One "step" = one pass of the whole path over one batch of synthetic sentences that is
already resident in HBM: encoder -> greedy decoder (fixed number of decoder steps) ->
postnet -> mel-to-linear -> 60-iteration Griffin-Lim -> de-emphasis.  Default workload:
64 sentences x 128 phoneme ids, 800 decoder steps, r = 1 (51 200 mel frames per step), the
"Tacotron2 batch=64 synthetic sentences, fixed 800 decoder steps" configuration.  With
--gpus N every rank runs that workload on its own GPU (independent utterances, no
data-path collective); value = frames of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      the decoder step (the dominant stage) against the chip's fp32-MFMA / HBM peak,
                from HIP-event timing on the launch stream inside this run;
  cpu_baseline  the CPU oracle (a restatement of the reference's PyTorch/NumPy path) timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N=1);
  batch1        batch-1 latency and real-time factor of the same path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak
DEC_WEIGHT_BYTES = 18928802 * 4  # decoder parameters read once per step (inputs_layer excluded)


def dec_step_bytes(B, T):
    return DEC_WEIGHT_BYTES + 2580.0 * T * B


def dec_step_flops(B, T):
    return B * (37.81e6 + 13440.0 * T)


def build_model(device, r):
    from tts_amd import synthetic
    from tts_amd.audio import AudioProcessor
    from tts_amd.tacotron2 import Tacotron2
    sd = synthetic.t2_state_dict(seed=0)
    model = Tacotron2(num_chars=129, num_speakers=0, r=7, attn_norm="sigmoid", prenet_type="bn",
                      prenet_dropout=False, separate_stopnet=True)
    model.load_state_dict(sd)
    model.cuda(device).eval()
    model.decoder.set_r(r)
    ap = AudioProcessor(sample_rate=22050, num_mels=80, min_level_db=-100, frame_shift_ms=12.5,
                        frame_length_ms=50, ref_level_db=20, num_freq=1025, power=1.5, preemphasis=0.98,
                        signal_norm=True, symmetric_norm=True, max_norm=4, mel_fmin=0.0, mel_fmax=8000.0,
                        clip_norm=True, griffin_lim_iters=60, verbose=False,
                        device=torch.device("cuda", device))
    return model, ap, sd


class Workload:
    def __init__(self, model, ap, ids, dec_steps, phase):
        self.model, self.ap, self.ids, self.dec_steps, self.phase = model, ap, ids, dec_steps, phase
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        self.stage_ms = np.zeros(5)
        self.n = 0

    def run(self, record=False):
        m, ap, ev = self.model, self.ap, self.ev
        if record:
            ev[0].record()
        enc = m.encode(self.ids)
        if record:
            ev[1].record()
        mel, align, stop, n_steps, steps = m.decode(enc, None, fixed_steps=self.dec_steps)
        if record:
            ev[2].record()
        post = m.postnet(mel)
        if record:
            ev[3].record()
        mag = ap.spec_to_mag(post)
        if record:
            ev[4].record()
        wav = ap.griffin_lim_batch(mag, self.phase)
        if record:
            ev[5].record()
            torch.cuda.synchronize()
            self.stage_ms += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(5)])
            self.n += 1
        return wav, mel

    def stages(self):
        names = ["encoder", "decoder", "postnet", "spec_to_mag", "griffin_lim"]
        return {k: float(v / max(self.n, 1)) for k, v in zip(names, self.stage_ms)}


def cpu_baseline(sd, T_in, sample_steps, r):
    """The CPU oracle on a bounded sample: one utterance, `sample_steps` decoder steps, then
    postnet + mel-to-linear + 60-iteration Griffin-Lim + de-emphasis for those frames."""
    from oracle import audio_oracle as ao
    from oracle import t2_oracle
    from tts_amd import synthetic
    ids = synthetic.t2_text_ids(1000, 1, T_in)
    orc_ap = ao.AudioOracle(**ao.T2_AUDIO)
    inv = orc_ap.inv_mel_basis()
    t0 = time.perf_counter()
    out = t2_oracle.inference(sd, ids, r=r, attn_norm="sigmoid", prenet_type="bn", max_steps=sample_steps,
                              fixed_steps=True)
    t1 = time.perf_counter()
    mel = out["postnet_output"][0].numpy().T
    ph = synthetic.gl_init_phase(1, 1025, mel.shape[1])
    wav = orc_ap.inv_mel_spectrogram(mel, ph, inv)
    t2 = time.perf_counter()
    frames = mel.shape[1]
    return dict(value=frames / (t2 - t0), unit="mel frames/s", cores=int(torch.get_num_threads()), kind="port",
                sample="1 utterance, T_in=%d, %d decoder steps (r=%d) + postnet + 60-iter Griffin-Lim; "
                       "model %.2fs, vocoder %.2fs; oracle = CPU restatement of the reference path "
                       "(torch CPU ops + numpy.fft)" % (T_in, sample_steps, r, t1 - t0, t2 - t1),
                model_s=t1 - t0, vocoder_s=t2 - t1, wav_rms=float(np.sqrt(np.mean(wav ** 2))))


def tacotron_v1_batch1(dev, T_in=100, steps=200):
    """The Tacotron (CBHG postnet, linear spectrogram) configuration at batch 1: embedding -> encoder CBHG ->
    greedy decoder -> post-CBHG + last_linear -> 60-iteration Griffin-Lim (win 1102, reflect padding)."""
    from tts_amd import synthetic
    from tts_amd.audio import AudioProcessor
    from tts_amd.tacotron import Tacotron
    sd = synthetic.t1_state_dict(seed=0, gains=dict(stop=0.0, stop_bias=-20.0, v=2.0))   # runs to max_decoder_steps
    m = Tacotron(61, 256, 1025, 80, 2, None, 5)
    m.load_state_dict(sd)
    m.cuda(dev.index).eval()
    m.decoder.max_decoder_steps = steps
    ap = AudioProcessor(sample_rate=22050, num_mels=80, min_level_db=-100, frame_shift_ms=12.5, frame_length_ms=50,
                        ref_level_db=20, num_freq=1025, power=1.5, preemphasis=0.98, signal_norm=True,
                        symmetric_norm=False, max_norm=1, clip_norm=True, griffin_lim_iters=60, tacotron_v1=True,
                        verbose=False, device=dev)
    ids = torch.as_tensor(synthetic.t1_text_ids(7, T_in)).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]

    def run(record=False):
        if record:
            ev[0].record()
        enc = m.encode(ids)
        if record:
            ev[1].record()
        dec = m.decode(enc)
        if record:
            ev[2].record()
        lin = m.postnet(dec[0])
        if record:
            ev[3].record()
        ph = torch.rand(1, lin.shape[1], 1025, device=dev)
        wav = ap.griffin_lim_batch(ap.spec_to_mag(lin, mel=False), ph)
        if record:
            ev[4].record()
        return lin.shape[1], int(dec[3].max())

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        frames, n_steps = run()
    torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 3
    run(record=True)
    torch.cuda.synchronize()
    st = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    audio_s = frames * 275 / 22050.0
    return {"workload": "Tacotron (v1) B=1, T_in=%d, %d decoder steps, r=2, post-CBHG + last_linear + 60-iter Griffin-Lim"
                        % (T_in, n_steps), "latency_s": lat, "audio_s": audio_s, "real_time_factor": lat / audio_s,
            "frames_per_s": frames / lat, "decoder_us_per_step": st[1] * 1e3 / n_steps,
            "stage_ms": {"encoder": st[0], "decoder": st[1], "postnet": st[2], "griffin_lim": st[3]}}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--t-in", type=int, default=128)
    p.add_argument("--dec-steps", type=int, default=800)
    p.add_argument("--r", type=int, default=1)
    p.add_argument("--cpu-sample-steps", type=int, default=1600)  # ~11 s of host work
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-batch1", action="store_true")
    a = p.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from tts_amd import synthetic
    model, ap, sd = build_model(local_rank, a.r)
    B, T, S, r = a.batch, a.t_in, a.dec_steps, a.r
    # every rank gets its own sentences (the 4096-utterance corpus dealt out in batches)
    ids = torch.as_tensor(synthetic.t2_text_ids(1234 + rank, B, T)).to(dev)
    F_total = S * r
    phase = torch.rand(B, F_total, 1025, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    wl = Workload(model, ap, ids, S, phase)

    from tts_amd import corpus
    # warm-up, then exactly `steps` timed passes between synchronise + barrier fences; max over ranks
    dt = corpus.timed_steps(wl.run, a.steps, a.warmup, dist, dev)
    ms_per_step = dt / a.steps * 1e3
    frames_per_step = B * F_total * world
    value = frames_per_step / (ms_per_step * 1e-3)

    # stage split and decoder-step roofline from HIP events on the launch stream (untimed extra passes)
    for _ in range(2):
        wl.run(record=True)
    st = wl.stages()
    dec_us = st["decoder"] * 1e3 / S
    flops, nbytes = dec_step_flops(B, T), dec_step_bytes(B, T)
    mfma_floor_us = flops / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e6
    hbm_floor_us = nbytes / (HBM_PEAK_GBS * 1e9) * 1e6
    if mfma_floor_us >= hbm_floor_us:
        roof = dict(bound="mfma", achieved=flops / (dec_us * 1e-6) / 1e12, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s")
    else:
        roof = dict(bound="hbm", achieved=nbytes / (dec_us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
    roof["frac"] = roof["achieved"] / roof["peak"]
    # HBM bytes per step from the PMC passes kept under profiles/ (a separate rocprofv3 run on the same
    # workload; counters cannot be read from inside this process)
    roof["traffic"] = None
    pmc_path = os.path.join(ROOT, "profiles", "r1_decoder_pmc.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        if (pmc.get("B"), pmc.get("T")) == (B, T):
            roof["traffic"] = pmc["hbm_bytes_per_step"]
            roof["traffic_source"] = "profiles/r1_decoder_pmc.csv"
    roof["kernel"] = ("t2 decoder step (5 launches: prenet-2, attention LSTM + query shares, fused attention, "
                      "decoder LSTM, projection; the early LSTM columns ride along as partial pieces)")
    roof["us_per_launch_group"] = dec_us
    roof["algorithmic_bytes"] = nbytes
    roof["algorithmic_flops"] = flops
    roof["hbm_equiv_gbs"] = nbytes / (dec_us * 1e-6) / 1e9

    out = {
        "metric": "mel frames/sec, Tacotron2+Griffin-Lim", "value": value, "unit": "mel frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded weights in the repo config's architecture, random phoneme ids, random GL phase)",
        "config": {"workload": "Tacotron2 batch=%d synthetic sentences (T_in=%d), fixed %d decoder steps, r=%d, "
                               "postnet + 60-iter Griffin-Lim, per GPU" % (B, T, S, r),
                   "global_batch": B * world, "parallelism": "dp%d (independent utterances, no collective)" % world},
        "real_time_factor_batch": ms_per_step * 1e-3 / (B * F_total * 275 / 22050.0),
        "stage_ms": st, "roofline": roof,
    }

    if rank == 0 and world == 1 and not a.no_batch1:
        ids1 = torch.as_tensor(synthetic.t2_text_ids(99, 1, 100)).to(dev)
        ph1 = torch.rand(1, F_total, 1025, device=dev)
        wl1 = Workload(model, ap, ids1, S, ph1)
        wl1.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            wl1.run()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 3
        wl1.run(record=True)
        s1 = wl1.stages()
        d_us = s1["decoder"] * 1e3 / S
        out["batch1"] = {"latency_s": lat, "audio_s": F_total * 275 / 22050.0,
                         "real_time_factor": lat / (F_total * 275 / 22050.0), "frames_per_s": F_total / lat,
                         "stage_ms": s1, "decoder_us_per_step": d_us,
                         "decoder_hbm_gbs": dec_step_bytes(1, 100) / (d_us * 1e-6) / 1e9,
                         "decoder_hbm_frac": dec_step_bytes(1, 100) / (d_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "workload": "B=1, T_in=100, %d decoder steps, r=%d" % (S, r)}
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):   # the model reports 'max_decoder_steps' like the reference
            v1 = tacotron_v1_batch1(dev)
        out["tacotron_v1_batch1"] = v1
        # long-form input: ~1000 symbols, 2000 decoder steps (attention over a long encoder axis)
        idsL = torch.as_tensor(synthetic.t2_text_ids(77, 1, 1000)).to(dev)
        wlL = Workload(model, ap, idsL, 2000, torch.rand(1, 2000 * r, 1025, device=dev))
        wlL.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wlL.run()
        torch.cuda.synchronize()
        latL = time.perf_counter() - t0
        wlL.run(record=True)
        sL = wlL.stages()
        out["longform"] = {"workload": "B=1, T_in=1000, 2000 decoder steps, r=%d" % r, "latency_s": latL,
                           "real_time_factor": latL / (2000 * r * 275 / 22050.0), "stage_ms": sL,
                           "decoder_us_per_step": sL["decoder"] * 1e3 / 2000}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd, T, a.cpu_sample_steps, r)
        # the same path on one host thread (a quarter of the sample), as a scalar reference point
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            one = cpu_baseline(sd, T, max(a.cpu_sample_steps // 4, 8), r)
        finally:
            torch.set_num_threads(nt)
        out["cpu_baseline"]["one_thread"] = {"value": one["value"], "unit": one["unit"], "cores": 1,
                                             "sample": one["sample"]}
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
