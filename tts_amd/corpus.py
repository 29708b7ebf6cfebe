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


def corpus_texts(lengths, seed=1234, num_chars=129):
    """Phoneme ids of the synthetic corpus: utterance i is the first lengths[i] entries of row i of one
    seeded table, so it does not depend on how the corpus is cut up."""
    lengths = np.asarray(lengths)
    table = synthetic.integers(seed, "corpus.ids", (len(lengths), int(lengths.max())), 3, num_chars)
    return [table[i, :int(n)] for i, n in enumerate(lengths)]


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


def timed_steps(run, steps, warmup, dist=None, device=None, own=False):
    """`warmup` untimed calls of run(), then exactly `steps` timed ones, bracketed by a device
    synchronise + barrier on both sides.  Returns the elapsed seconds, MAX over ranks; with `own`
    a pair (that, this rank's own busy seconds up to its device synchronise, before the barrier)."""
    cuda = device is not None and torch.device(device).type == "cuda"

    def sync():
        if cuda:
            torch.cuda.synchronize(device)

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    for _ in range(warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    sync()
    busy = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    dt = reduce_scalar(dt, "max", dist, device)
    return (dt, busy) if own else dt


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


class StagedCorpus:
    """This rank's share of a corpus, resident in device memory before any timing starts: padded ids,
    text lengths, decoder steps and Griffin-Lim initial phases of every batch.

    Decoder steps are fixed by the text: utterance i gets min(max_steps, steps_per_id * len_i) steps (r frames
    each); a batch runs to its longest member and the shorter ones are cut to their own count afterwards
    (postnet and vocoder take per-utterance frame counts), so an utterance's frames are those of its
    batch-1 run.  The plan (which utterances share a batch) does not depend on world or rank; only the
    dealing of batches to ranks does.  A batch's initial phases come from a device generator seeded with the
    batch's index in the plan."""

    def __init__(self, model, ap, lengths, texts, batch=64, world=1, rank=0, policy="lpt", steps_per_id=6,
                 max_steps=3000, phase_seed=0, device=None):
        self.model, self.ap = model, ap
        self.r = model.decoder.r
        lengths = np.asarray(lengths)
        batches = plan_batches(lengths, batch)
        self.items = []
        self.frames = 0          # frames that count: sum over utterances of their own steps * r
        self.frames_padded = 0   # frames the decoder computes: rows * steps of the batch * r
        self.n_utts = 0
        dev = device if device is not None else model.device
        for bi in deal(batches, lengths, world, policy)[rank]:
            idx = batches[bi]
            ids, lens = pad_batch(texts, idx)
            steps_b = np.minimum(max_steps, steps_per_id * lens)
            steps = int(steps_b.max())
            n_frames = (steps_b * self.r).astype(np.int32)
            gen = torch.Generator(device=dev).manual_seed(phase_seed + int(bi))
            phase = torch.rand(len(idx), steps * self.r, 1025, device=dev, generator=gen)
            self.items.append(dict(batch_index=int(bi), idx=idx, ids=torch.as_tensor(ids).to(dev),
                                   lens=torch.as_tensor(lens.astype(np.int32)).to(dev), steps=steps,
                                   n_frames=torch.as_tensor(n_frames).to(dev), phase=phase))
            self.frames += int(n_frames.sum())
            self.frames_padded += len(idx) * steps * self.r
            self.n_utts += len(idx)

    def run(self, keep=False):
        """One pass over the shard: encoder, decoder, postnet, mel-to-linear, Griffin-Lim per batch.
        With `keep` returns {utterance index: (mel [F,80], wav [hop*(F-1)])} on the host."""
        m, ap = self.model, self.ap
        out = {}
        for it in self.items:
            enc = m.encode(it["ids"], it["lens"])
            mel, align, stop, n_steps, steps = m.decode(enc, it["lens"], fixed_steps=it["steps"])
            post = m.postnet(mel, it["n_frames"])
            wav = ap.griffin_lim_batch(ap.spec_to_mag(post), it["phase"], it["n_frames"])
            if keep:
                nf = it["n_frames"].cpu().numpy()
                post_h, wav_h = post.cpu().numpy(), wav.cpu().numpy()
                for row, i in enumerate(it["idx"]):
                    F = int(nf[row])
                    out[int(i)] = (post_h[row, :F].copy(), wav_h[row, :ap.hop_length * max(F - 1, 0)].copy())
        return out if keep else None


def gather_rank_stats(values, dist=None, device=None):
    """[world, len(values)] float64 array with every rank's `values` (reporting only: frame counts and busy
    times, a few numbers per rank)."""
    v = torch.tensor([float(x) for x in values], dtype=torch.float64, device=device if device is not None else "cpu")
    if dist is None:
        return v.cpu().numpy()[None]
    bufs = [torch.empty_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, v)
    return torch.stack(bufs).cpu().numpy()
