#!/usr/bin/env python3
"""Benchmark of the synthesis hot path: mel frames/s of Tacotron2 + Griffin-Lim.

One "step" = one pass of the whole path over one batch of synthetic sentences that is
already resident in HBM: encoder -> greedy decoder -> postnet -> mel-to-linear -> 60-iteration
Griffin-Lim -> de-emphasis.

Workloads (--workload):
  batch   (default) 64 sentences x 128 phoneme ids, 800 fixed decoder steps, r = 1 (51 200 mel frames
          per step): the "Tacotron2 batch=64 synthetic sentences, fixed 800 decoder steps" configuration.
          With --gpus N every rank runs it on its own GPU with its own sentences; value = frames of all
          ranks / max-over-ranks time ("weak" scaling).
  corpus  the 4096-utterance ragged corpus (text lengths uniform in 20..200, seed 1234; min(3000, 6 x length)
          decoder steps per utterance), length-sorted into batches of 64 that are dealt to the ranks
          longest-processing-time first; a step = one pass over the rank's share; value = frames of the
          whole corpus / max-over-ranks time ("strong" scaling), with per-rank frames and busy time.
The default line also carries a `corpus` block: the whole corpus over the N ranks when N > 1, one eighth of
it (what one of eight GPUs would get) when N = 1.

--gpus N without a torch.distributed environment starts the N ranks itself (fresh child processes, before
anything in this process touches the GPU); under `torch.distributed.run` the ranks are already there and
--gpus must equal WORLD_SIZE.  Independent utterances, weights replicated, no data-path collective:
torch.distributed (RCCL) carries the timing barrier and a few reporting scalars only.

The JSON line also carries
  roofline      the decoder step (the dominant stage) against the chip's fp32-MFMA / HBM peak,
                from HIP-event timing on the launch stream inside this run;
  roofline_gl   the Griffin-Lim stage against the HBM and vector-ALU peaks;
  roofline_postnet / roofline_encoder   the conv stacks (and the encoder BiLSTM) against the fp32-MFMA peak;
  decoder_by_batch  the decoder step at batch 128 and 256 (the corpus workload's batch) against the same peak;
  cpu_baseline  the CPU oracle (a restatement of the reference's PyTorch/NumPy path) timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N=1);
  batch1        batch-1 latency and real-time factor of the same path, fixed 800 steps;
  batch1_stop_driven  the same at batch 1 with the stop token ending the decode.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (= fp32 vector peak)
DEC_WEIGHT_BYTES = 18928802 * 4  # decoder parameters read once per step (inputs_layer excluded)
GL_BYTES_PER_FRAME = 61 * 1025 * 4 + 121 * 275 * 4   # 60 iterations: magnitudes read 61 x, signal read + written
GL_FLOPS_PER_FRAME = 121 * 2.5 * 2048 * 11           # 121 real FFTs of 2048 points


# One exchange between the workgroups of a persistent team (every workgroup publishes its words with one store,
# every consumer polls each word once and hands it on through LDS): 1.2 us from the publishing store to the
# consumers' barrier, the mean of the five on a Tacotron2 step in the kernel's own timestamps (0.9 - 1.5 us;
# tools/t2_team_timing.py; tools/exchange_probe.hip: 1.15 us bare at 32 workgroups / 256 words).  The serial chain
# of a Tacotron2 step holds five of them (x1 | x2 | h_a | partial energies | h_d), that of a Tacotron (v1) step six.
TEAM_EXCHANGE_US = 1.2
TEAM_EXCHANGES_T2 = 5
TEAM_EXCHANGES_T1 = 6


def team_bound(us_per_step, exchanges):
    floor = exchanges * TEAM_EXCHANGE_US
    return {"bound": "exchange latency", "exchanges_on_the_serial_chain": exchanges,
            "exchange_us": TEAM_EXCHANGE_US, "floor_us": floor, "frac": floor / us_per_step}


def team_traffic(case):
    """HBM bytes per decoder step of a team kernel from the counter passes kept under profiles/ (a separate
    rocprofv3 run: tools/profile_team.sh), or None."""
    path = os.path.join(ROOT, "profiles", "r3_team_pmc.csv")
    if not os.path.exists(path):
        return None
    import csv
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["case"] == case and "team_kernel" in row["kernel"] and "pack" not in row["kernel"]:
                return float(row["hbm_bytes_per_step"])
    return None


def dec_step_bytes(B, T):
    return DEC_WEIGHT_BYTES + 2580.0 * T * B


def dec_step_flops(B, T):
    return B * (37.81e6 + 13440.0 * T)


def build_model(device, r, gains=None):
    from tts_amd import synthetic
    from tts_amd.audio import AudioProcessor
    from tts_amd.tacotron2 import Tacotron2
    sd = synthetic.t2_state_dict(seed=0, gains=gains)
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


def cpu_config1():
    """The CPU oracle on the stop-driven batch-1 request of `batch1_stop_driven` (T_in = 100, the late-stop
    checkpoint, about 308 decoder steps): latency end to end with the model / vocoder split."""
    from oracle import audio_oracle as ao
    from oracle import t2_oracle
    from tts_amd import synthetic
    sd = synthetic.t2_state_dict(seed=0, prenet_type="bn", gains=synthetic.T2_LATE_STOP_GAINS)
    ids = synthetic.t2_text_ids(99, 1, 100)
    orc_ap = ao.AudioOracle(**ao.T2_AUDIO)
    inv = orc_ap.inv_mel_basis()
    t0 = time.perf_counter()
    out = t2_oracle.inference(sd, ids, r=1, attn_norm="sigmoid", prenet_type="bn", max_steps=3000)
    t1 = time.perf_counter()
    mel = out["postnet_output"][0].numpy().T
    wav = orc_ap.inv_mel_spectrogram(mel, synthetic.gl_init_phase(1, 1025, mel.shape[1]), inv)
    t2 = time.perf_counter()
    n = int(out["n_steps"][0])
    return dict(workload="B=1, T_in=100, stop-driven (threshold 0.7), r=1: encoder + decoder + postnet, then 60-iter "
                         "Griffin-Lim", n_steps=n, latency_s=t2 - t0, model_s=t1 - t0, vocoder_s=t2 - t1,
                frames_per_s=mel.shape[1] / (t2 - t0), cores=int(torch.get_num_threads()), kind="port",
                wav_rms=float(np.sqrt(np.mean(wav ** 2))))


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


def batch1_stop_driven(dev, ap):
    """Batch 1 as a server request runs it: the decoder stops when its stop token exceeds 0.7 (a late-stop
    synthetic checkpoint: the stop step depends on the text and lies near step 309), then postnet and
    Griffin-Lim over the frames it produced."""
    from tts_amd import synthetic
    model, _, _ = build_model(dev.index, 1, gains=synthetic.T2_LATE_STOP_GAINS)
    model.decoder.max_decoder_steps = 3000
    ids = torch.as_tensor(synthetic.t2_text_ids(99, 1, 100)).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def run(record=False):
        enc = model.encode(ids)
        if record:
            ev[0].record()
        mel, align, stop, n_steps, steps = model.decode(enc, None)
        if record:
            ev[1].record()
        post = model.postnet(mel)
        ph = torch.rand(1, post.shape[1], 1025, device=dev)
        wav = ap.griffin_lim_batch(ap.spec_to_mag(post), ph)
        if record:
            ev[2].record()
        return int(n_steps[0]), model.last_steps_launched()

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            n, launched = run()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 3
        run(record=True)
        torch.cuda.synchronize()
    dec_ms, rest_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    audio_s = n * 275 / 22050.0
    return {"workload": "B=1, T_in=100, stop-driven (threshold 0.7, max_decoder_steps 3000), r=1",
            "n_steps": n, "steps_launched": launched, "latency_s": lat, "audio_s": audio_s,
            "real_time_factor": lat / audio_s, "frames_per_s": n / lat,
            "decoder_ms": dec_ms, "decoder_us_per_needed_step": dec_ms * 1e3 / n, "postnet_vocoder_ms": rest_ms}


# ---------------------------------------------------------------------------------------------- ranks
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def kfd_gpu_nodes(base="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs the kernel driver lists, read from sysfs (topology nodes with SIMDs; CPU nodes have none) — no HIP or
    torch.cuda call, so a process that only asks this has not opened the GPU.  None when the topology is absent."""
    if not os.path.isdir(base):
        return None
    n = 0
    for node in os.listdir(base):
        try:
            with open(os.path.join(base, node, "properties")) as f:
                for ln in f:
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError, IndexError):
            pass
    return n


def launch_ranks(a, argv):
    """Start a.gpus fresh child processes, one rank each (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), wait
    for them and return the worst exit status.  This process never touches the GPU (no HIP call, no torch.cuda
    call): the node's GPU count comes from sysfs, and each rank checks for itself that its device exists.  When
    a rank fails, the others — which would wait for it at the rendezvous — are ended.  Rank 0's JSON line goes
    to this process's stdout."""
    if not a.dry_run:
        n_dev = kfd_gpu_nodes()
        if n_dev is not None and a.gpus > n_dev:
            sys.stderr.write("bench.py: --gpus %d asked for but this node shows %d GPU(s)\n" % (a.gpus, n_dev))
            return 2
    port = _free_port()
    procs = []
    for rank in range(a.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    worst, live = 0, list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and worst == 0:
                worst = rc if rc > 0 else 1
                for q in live:         # exactly the children started above
                    q.terminate()
    return worst


def corpus_plan_summary(a, world):
    from tts_amd import corpus
    lengths = corpus.corpus_lengths(a.corpus_utts, seed=1234)
    batches = corpus.plan_batches(lengths, a.corpus_batch)
    return lengths, batches, corpus.deal(batches, lengths, world, "lpt")


def dry_run(a, rank, world):
    """The launcher and the corpus plan without a GPU: every rank derives its shard, the reporting exchange
    runs over gloo, rank 0 prints what each rank would synthesise."""
    from tts_amd import corpus
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths, batches, dealt = corpus_plan_summary(a, world)
    mine = [batches[i] for i in dealt[rank]]
    utts = sorted(int(i) for b in mine for i in b)
    frames = float(sum(int(np.minimum(3000, 6 * lengths[b]).sum()) * a.r for b in mine))
    stats = corpus.gather_rank_stats([rank, len(mine), len(utts), frames, utts[0], utts[-1],
                                      float(np.sum(utts) % 1000003)], dist)
    total = corpus.reduce_scalar(frames, "sum", dist)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "workload": a.workload, "corpus_utterances": a.corpus_utts,
                          "total_frames": total,
                          "ranks": [dict(rank=int(s[0]), batches=int(s[1]), utterances=int(s[2]), frames=s[3],
                                         first_utt=int(s[4]), last_utt=int(s[5]), utt_checksum=int(s[6]))
                                    for s in stats]}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_corpus(a, model, ap, rank, world, dist, dev, n_utts, shard_of=None):
    """Time `a.steps` passes over this rank's share of the corpus.  shard_of = (world', rank'): take the share
    rank' of world' ranks would get (used at N=1 to run what one of eight GPUs would)."""
    from tts_amd import corpus
    lengths = corpus.corpus_lengths(n_utts, seed=1234)
    texts = corpus.corpus_texts(lengths, seed=1234)
    w, r_ = (world, rank) if shard_of is None else shard_of
    staged = corpus.StagedCorpus(model, ap, lengths, texts, batch=a.corpus_batch, world=w, rank=r_, policy="lpt",
                                 steps_per_id=6, max_steps=3000, device=dev)
    dt, busy = corpus.timed_steps(staged.run, a.steps, a.warmup, dist, dev, own=True)
    stats = corpus.gather_rank_stats([staged.frames, staged.frames_padded, busy / a.steps, staged.n_utts,
                                      len(staged.items)], dist, dev if dist is not None else None)
    frames = float(stats[:, 0].sum())
    ms = dt / a.steps * 1e3
    return {"workload": "Tacotron2, %d-utterance synthetic corpus (T_in uniform 20..200, seed 1234), min(3000, 6 x T_in) "
                        "decoder steps, r=%d, length-sorted batches of %d dealt longest-processing-time first%s, "
                        "postnet + 60-iter Griffin-Lim" % (n_utts, staged.r, a.corpus_batch,
                                                            "" if shard_of is None else
                                                            "; share %d of %d" % (shard_of[1], shard_of[0])),
            "frames": frames, "ms_per_pass": ms, "frames_per_s": frames / (ms * 1e-3),
            "per_rank": [dict(frames=float(s[0]), frames_computed=float(s[1]), busy_ms=float(s[2] * 1e3),
                              utterances=int(s[3]), batches=int(s[4])) for s in stats],
            "imbalance": float(stats[:, 2].max() / max(stats[:, 2].mean(), 1e-12))}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", choices=("batch", "corpus"), default="batch")
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--t-in", type=int, default=128)
    p.add_argument("--dec-steps", type=int, default=800)
    p.add_argument("--r", type=int, default=1)
    p.add_argument("--corpus-utts", type=int, default=4096)
    # utterances per length-sorted batch of the corpus workload (measured on one GPU, whole corpus: 64 -> 840 k
    # frames/s, 128 -> 999 k, 256 -> 1.12 M: weights and launch latencies are shared by more columns; the
    # longest-processing-time-first deal of the 16 batches over 8 ranks stays within 0.5 % of balance)
    p.add_argument("--corpus-batch", type=int, default=256)
    p.add_argument("--cpu-sample-steps", type=int, default=1600)  # ~11 s of host work
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-batch1", action="store_true")
    p.add_argument("--no-corpus", action="store_true")
    p.add_argument("--dry-run", action="store_true", help="launcher + corpus plan only (no GPU; gloo)")
    a = p.parse_args()
    if a.gpus < 1:
        p.error("--gpus must be >= 1")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE=%d\n" % (a.gpus, world))
        sys.exit(2)
    if a.dry_run:
        sys.exit(dry_run(a, rank, world))
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d has no GPU (%d visible)\n" % (local_rank, torch.cuda.device_count()))
        sys.exit(2)

    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from tts_amd import corpus, synthetic
    model, ap, sd = build_model(local_rank, a.r)
    B, T, S, r = a.batch, a.t_in, a.dec_steps, a.r
    F_total = S * r
    out = {"metric": "mel frames/sec, Tacotron2+Griffin-Lim", "unit": "mel frames/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "vs_baseline": None, "dtype": "f32",
           "data": "synthetic (seeded weights in the repo config's architecture, random phoneme ids, random GL phase)"}

    if a.workload == "corpus":
        c = run_corpus(a, model, ap, rank, world, dist, dev, a.corpus_utts)
        out.update(value=c["frames_per_s"], ms_per_step=c["ms_per_pass"], scaling="strong",
                   config={"workload": c["workload"], "global_batch": a.corpus_utts,
                           "parallelism": "dp%d (independent utterances, no collective)" % world},
                   corpus=c)
    # the fixed batch: headline of the default workload, and the source of the stage split / roofline
    # every rank gets its own sentences
    ids = torch.as_tensor(synthetic.t2_text_ids(1234 + rank, B, T)).to(dev)
    phase = torch.rand(B, F_total, 1025, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    wl = Workload(model, ap, ids, S, phase)
    if a.workload == "batch":
        # warm-up, then exactly `steps` timed passes between synchronise + barrier fences; max over ranks
        dt = corpus.timed_steps(wl.run, a.steps, a.warmup, dist, dev)
        ms_per_step = dt / a.steps * 1e3
        out.update(value=B * F_total * world / (ms_per_step * 1e-3), ms_per_step=ms_per_step, scaling="weak",
                   config={"workload": "Tacotron2 batch=%d synthetic sentences (T_in=%d), fixed %d decoder steps, r=%d, "
                                       "postnet + 60-iter Griffin-Lim, per GPU" % (B, T, S, r),
                           "global_batch": B * world,
                           "parallelism": "dp%d (independent utterances, no collective)" % world},
                   real_time_factor_batch=ms_per_step * 1e-3 / (B * F_total * 275 / 22050.0))
        if not a.no_corpus:
            # the ragged corpus next to it: all of it over the N ranks, or one eighth of it on a single GPU
            out["corpus"] = run_corpus(a, model, ap, rank, world, dist, dev, a.corpus_utts,
                                       shard_of=(8, 0) if world == 1 else None)
    else:
        wl.run()

    # stage split and decoder-step roofline from HIP events on the launch stream (untimed extra passes)
    # (each recorded pass is enqueued behind an unrecorded one that is still executing, so that the events bracket
    # the GPU time of a stage and not the host's launch rate on an empty queue - the timed loop above runs the
    # same way, the host ahead of the device)
    for _ in range(2):
        wl.run()
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
    for name in ("r3_decoder_pmc.json", "r2_decoder_pmc.json", "r1_decoder_pmc.json"):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if (pmc.get("B"), pmc.get("T")) == (B, T):
                roof["traffic"] = pmc["hbm_bytes_per_step"]
                roof["traffic_source"] = "profiles/" + name.replace(".json", ".csv")
                break
    roof["kernel"] = ("t2 decoder step (launch group: prenet-2, attention LSTM + query shares, fused attention, "
                      "decoder LSTM, projection; the early LSTM columns ride along as partial pieces)")
    roof["us_per_launch_group"] = dec_us
    roof["algorithmic_bytes"] = nbytes
    roof["algorithmic_flops"] = flops
    roof["hbm_equiv_gbs"] = nbytes / (dec_us * 1e-6) / 1e9
    gl_s = st["griffin_lim"] * 1e-3
    n_fr = B * F_total
    out.update(stage_ms=st, roofline=roof,
               roofline_gl={"kernel": "gl_fused_kernel (one launch per Griffin-Lim iteration) + de-emphasis",
                            "frames": n_fr, "ms": st["griffin_lim"],
                            "hbm": {"achieved": n_fr * GL_BYTES_PER_FRAME / gl_s / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": n_fr * GL_BYTES_PER_FRAME / gl_s / 1e9 / HBM_PEAK_GBS},
                            "valu": {"achieved": n_fr * GL_FLOPS_PER_FRAME / gl_s / 1e12,
                                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": n_fr * GL_FLOPS_PER_FRAME / gl_s / 1e12 / MFMA_F32_PEAK_TFLOPS}})

    # the two conv stacks on the MFMA conv engine: postnet 80 -> 512 -> 512 -> 512 -> 512 -> 80 and the three
    # 512 -> 512 encoder blocks, five taps each (2 flops per multiply-add); the encoder stage also holds the BiLSTM
    post_flops = n_fr * 2.0 * 5 * (80 * 512 + 3 * 512 * 512 + 512 * 80)
    enc_conv_flops = B * T * 2.0 * 5 * 3 * 512 * 512
    enc_lstm_flops = B * T * 2.0 * 2 * 1024 * (512 + 256)
    out["roofline_postnet"] = {"kernel": "convmfma::conv_mfma_kernel x 5 (implicit GEMM, v_mfma_f32_32x32x2_f32)",
                               "bound": "mfma", "achieved": post_flops / (st["postnet"] * 1e-3) / 1e12,
                               "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": post_flops / (st["postnet"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                               "algorithmic_flops": post_flops, "ms": st["postnet"]}
    out["roofline_encoder"] = {"kernel": "3 conv blocks (MFMA conv engine) + BiLSTM (T_in dependent step launches)",
                               "bound": "latency (T_in serial BiLSTM steps)",
                               "achieved": (enc_conv_flops + enc_lstm_flops) / (st["encoder"] * 1e-3) / 1e12,
                               "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": (enc_conv_flops + enc_lstm_flops) / (st["encoder"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                               "algorithmic_flops": enc_conv_flops + enc_lstm_flops, "ms": st["encoder"]}

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
                         "workload": "B=1, T_in=100, %d decoder steps, r=%d" % (S, r),
                         "decoder_path": "persistent team kernel: one launch per run, every decoder weight resident "
                                         "in registers and LDS; activations cross the chip as tagged words",
                         # What bounds a step of the team kernel is not memory bandwidth (no weight byte moves) but
                         # the six dependent exchanges on its serial chain (x1 | x2 | h_a | partial energies |
                         # context | h_d), each one publish + poll round trip through the memory side.
                         "decoder_bound": team_bound(d_us, TEAM_EXCHANGES_T2),
                         # side note: the same step priced at the launch chain's algorithmic bytes (weights once per
                         # step); the counters show that these bytes do not move
                         "decoder_hbm_equivalent_gbs": dec_step_bytes(1, 100) / (d_us * 1e-6) / 1e9,
                         "decoder_hbm_bytes_per_step_measured": team_traffic("t2_100_800")}
        # the same decode as a chain of launches (what batch 2 and up, and texts beyond 128 symbols, run)
        model.set_team(0)
        wlc = Workload(model, ap, ids1, S, ph1)
        wlc.run()
        wlc.run(record=True)
        out["batch1"]["launch_chain_decoder_us_per_step"] = wlc.stages()["decoder"] * 1e3 / S
        model.set_team(-1)
        out["batch1_stop_driven"] = batch1_stop_driven(dev, ap)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):   # the model reports 'max_decoder_steps' like the reference
            v1 = tacotron_v1_batch1(dev)
        v1["decoder_bound"] = team_bound(v1["decoder_us_per_step"], TEAM_EXCHANGES_T1)
        v1["decoder_hbm_bytes_per_step_measured"] = team_traffic("t1_100_200")
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
                           "decoder_us_per_step": sL["decoder"] * 1e3 / 2000,
                           "decoder_bound": team_bound(sL["decoder"] * 1e3 / 2000, TEAM_EXCHANGES_T2),
                           "decoder_hbm_bytes_per_step_measured": team_traffic("t2_1000_2000")}
        # the same decoder at the batch sizes the corpus workload runs: marginal time per step between a 200-step and
        # a 100-step run (set-up cancels), against the fp32-MFMA peak.  The step of the headline batch (64 columns) is
        # bound by its chain of dependent launches; wider batches amortise it.
        sweep = []
        for Bs in (128, 256):
            ids_s = torch.as_tensor(synthetic.t2_text_ids(500 + Bs, Bs, T)).to(dev)
            enc_s = model.encode(ids_s)

            def dec_time(steps, reps=2):
                model.decode(enc_s, None, fixed_steps=steps)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    model.decode(enc_s, None, fixed_steps=steps)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps
            us = (dec_time(200) - dec_time(100)) / 100 * 1e6
            tf = dec_step_flops(Bs, T) / (us * 1e-6) / 1e12
            sweep.append({"batch": Bs, "T_in": T, "us_per_step": us, "achieved": tf, "unit": "TFLOP/s",
                          "frac": tf / MFMA_F32_PEAK_TFLOPS})
            del ids_s, enc_s
        out["decoder_by_batch"] = sweep
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
        # the batch-1 request of `batch1_stop_driven` on the host cores (latency, not a throughput sample)
        out["cpu_baseline"]["config1_stop_driven"] = cpu_config1()
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
