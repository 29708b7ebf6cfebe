"""Sharding a corpus of utterances over GPUs, one process per GPU, no data-path collective.

Utterances are independent (the reference synthesises the sentences of a request one after the
other with nothing shared but the read-only weights, Tacotron2/TTS/synthesizer.py:147-170;
Tacotron/TTS/synthesizer.py:57-78), so the only cross-rank traffic is two scalars for reporting:
the frame count (sum) and the elapsed time (max).  Placement never changes an utterance's
result: every batch row is computed as its batch-1 run would be.

`plan_batches` + `deal` are pure functions of the lengths, so every rank derives the same plan
without talking to the others.
"""
import time

import numpy as np
import torch

from . import synthetic


def corpus_lengths(n=4096, seed=1234, lo=20, hi=200):
    """Text lengths of the synthetic corpus: n utterances, T_in uniform in [lo, hi]."""
    return synthetic.integers(seed, "corpus.lengths", (n,), lo, hi + 1).astype(np.int64)


def plan_batches(lengths, batch=64):
    """Sort by length (longest first, index as tie-break so the plan is unique) and cut into
    batches: rows of one batch have similar decode lengths, so little of a batch idles."""
    lengths = np.asarray(lengths)
    order = np.lexsort((np.arange(len(lengths)), -lengths))
    return [order[i:i + batch] for i in range(0, len(order), batch)]


def deal(batches, lengths, world, policy="lpt"):
    """Assign batches to ranks.  'round_robin': batch i -> rank i % world.  'lpt' (longest
    processing time first): each batch, longest first, goes to the least-loaded rank; cost =
    rows x longest text (decode steps scale with the text length).  Returns a list of `world`
    lists of batch indices."""
    lengths = np.asarray(lengths)
    if world < 1:
        raise ValueError("world must be >= 1")
    out = [[] for _ in range(world)]
    if policy == "round_robin":
        for i in range(len(batches)):
            out[i % world].append(i)
        return out
    if policy != "lpt":
        raise ValueError("unknown policy %r" % (policy,))
    cost = [int(len(b) * lengths[b].max()) for b in batches]
    load = [0] * world
    for i in sorted(range(len(batches)), key=lambda i: (-cost[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        out[r].append(i)
        load[r] += cost[i]
    return out


def rank_batches(lengths, batch, world, rank, policy="lpt"):
    """The batches (arrays of utterance indices) that `rank` synthesises."""
    batches = plan_batches(lengths, batch)
    return [batches[i] for i in deal(batches, lengths, world, policy)[rank]]


def pad_batch(texts, idx):
    """ids [B, Tmax] int64 (zero padded) and lengths [B] for the utterances `idx`."""
    lens = np.array([len(texts[i]) for i in idx], dtype=np.int64)
    ids = np.zeros((len(idx), int(lens.max())), dtype=np.int64)
    for row, i in enumerate(idx):
        ids[row, :lens[row]] = texts[i]
    return ids, lens


def timed_steps(run, steps, warmup, dist=None, device=None):
    """`warmup` untimed calls of run(), then exactly `steps` timed ones, bracketed by a device
    synchronise + barrier on both sides.  Returns the elapsed seconds, MAX over ranks."""
    cuda = device is not None and torch.device(device).type == "cuda"

    def fence():
        if cuda:
            torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            if cuda:
                torch.cuda.synchronize(device)

    for _ in range(warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    fence()
    dt = time.perf_counter() - t0
    return reduce_scalar(dt, "max", dist, device)


def reduce_scalar(x, op, dist=None, device=None):
    """Sum or max of one float over ranks (the only cross-rank exchange of the path)."""
    if dist is None:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def synthesize_corpus(model, ap, texts, batch=64, world=1, rank=0, policy="lpt", init_phase_seed=0,
                      max_decoder_steps=None):
    """Synthesise this rank's share of `texts` (a list of int id sequences) with Tacotron2 +
    Griffin-Lim.  Returns {utterance index: dict(wav, mel, n_frames)}; the union over ranks covers
    every utterance once, and an utterance's entry does not depend on world, rank or policy."""
    lengths = np.array([len(t) for t in texts])
    if max_decoder_steps is not None:
        model.decoder.max_decoder_steps = int(max_decoder_steps)
    r = model.decoder.r
    out = {}
    for idx in rank_batches(lengths, batch, world, rank, policy):
        ids, lens = pad_batch(texts, idx)
        res = model.inference_batch(torch.as_tensor(ids), text_lengths=torch.as_tensor(lens))
        post = res["postnet_output"]
        n_frames = (res["n_steps"] * r).to(torch.int32)
        B, Fm, _ = post.shape
        phase = torch.zeros(B, Fm, 1025, device=post.device)
        nf = n_frames.cpu().numpy()
        for row, i in enumerate(idx):
            ph = synthetic.gl_init_phase(init_phase_seed + int(i), 1025, int(nf[row]))
            phase[row, :nf[row]] = torch.from_numpy(np.ascontiguousarray(ph.T.astype(np.float32)))
        mag = ap.spec_to_mag(post)
        wav = ap.griffin_lim_batch(mag, phase, n_frames).cpu().numpy()
        post = post.cpu().numpy()
        for row, i in enumerate(idx):
            F = int(nf[row])
            out[int(i)] = dict(wav=wav[row, :ap.hop_length * max(F - 1, 0)], mel=post[row, :F], n_frames=F)
    return out
