"""Device-side tail of a request (endpoint trim, gaps, int16 scaling) and the batched request path,
against the CPU oracle / the per-sentence path.  int16 samples are compared exactly."""
import io

import numpy as np
import pytest
import scipy.io.wavfile
import torch

pytestmark = pytest.mark.gpu


def _aps():
    from oracle import audio_oracle as ao
    from tts_amd.audio import AudioProcessor
    cfg = dict(ao.T2_AUDIO, griffin_lim_iters=4)
    orc = ao.AudioOracle(**cfg)
    ap = AudioProcessor(verbose=False, **{k: v for k, v in cfg.items() if k not in ("pad_mode", "win_from_hop")})
    return orc, ap


def _ref_tail(orc, wavs, gap, trim):
    parts, ends = [], []
    for w in wavs:
        e = orc.find_endpoint(w) if trim else len(w)
        ends.append(e)
        parts += [w[:e].astype(np.float64), np.zeros(gap)]
    cat = np.concatenate(parts)
    pcm = (cat * (32767 / max(0.01, np.max(np.abs(cat))))).astype(np.int16)
    return pcm, ends


@pytest.mark.parametrize("trim", [True, False])
def test_request_tail_matches_oracle(trim):
    orc, ap = _aps()
    rng = np.random.default_rng(5)
    L = 90000
    lens = [90000, 52000, 9000, 70000]          # 9000 < window: no candidate position at all
    speech = [30000, 20000, 9000, 70000]        # loud part; the rest is a faint (negative-going) tail
    wav = np.zeros((4, L), dtype=np.float32)
    for b in range(4):
        wav[b, :speech[b]] = rng.standard_normal(speech[b]) * 0.2
        wav[b, speech[b]:lens[b]] = -np.abs(rng.standard_normal(lens[b] - speech[b])) * 0.004
        wav[b, lens[b]:] = 7.0                   # beyond n_samples: must never be looked at
    wav[1, 36000] = 0.5                          # an isolated click postpones the endpoint of utterance 1
    pcm, ends, peak = ap.request_tail(torch.from_numpy(wav).cuda(), lens, gap=10000, trim=trim)
    ref_pcm, ref_ends = _ref_tail(orc, [wav[b, :lens[b]] for b in range(4)], 10000, trim)
    assert ends.tolist() == ref_ends
    if trim:
        assert ref_ends[0] < lens[0] and ref_ends[1] > 36000 and ref_ends[2] == 9000 and ref_ends[3] == 70000
    assert pcm.dtype == np.int16 and pcm.shape == ref_pcm.shape
    assert np.array_equal(pcm, ref_pcm)
    assert abs(peak - max(np.abs(wav[b, :ref_ends[b]]).max() for b in range(4))) == 0.0


def test_request_tail_quiet_signal_and_errors():
    orc, ap = _aps()
    wav = (np.random.default_rng(1).standard_normal((1, 30000)) * 1e-3).astype(np.float32)   # peak < 0.01
    pcm, ends, peak = ap.request_tail(torch.from_numpy(wav).cuda(), None, gap=5, trim=False)
    ref_pcm, _ = _ref_tail(orc, [wav[0]], 5, False)
    assert np.array_equal(pcm, ref_pcm) and peak < 0.01
    from tts_amd import _lib
    lib = _lib.load()
    with pytest.raises(ValueError):
        _lib.check(lib.ttsx_request_tail(None, None, 1, 10, 4, 1, 0.01, 0, 1, None, None, None, 0, None, None, None),
                   "ttsx_request_tail")


def test_batched_request_equals_sentence_by_sentence():
    """Three sentences of different lengths as one padded batch: same endpoints and, up to the
    fp32 differences between the batch and batch-1 engines, the same samples as the reference-shaped loop."""
    from tts_amd import synthetic
    from tts_amd.config import tacotron2_repo_config
    from tts_amd.synthesis import Synthesizer
    from tts_amd.tacotron2 import Tacotron2
    orc, ap = _aps()
    sd = synthetic.t2_state_dict(seed=0)
    m = Tacotron2(num_chars=129, num_speakers=0, r=7, attn_norm="sigmoid", prenet_type="bn", prenet_dropout=False,
                  separate_stopnet=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    m.decoder.max_decoder_steps = 24
    m.decoder.set_r(7)
    syn = Synthesizer(m, ap, tacotron2_repo_config())
    lens = [30, 21, 26]
    sents = [synthetic.t2_text_ids(700 + i, 1, T)[0].tolist() for i, T in enumerate(lens)]
    # phases sized for the longest possible output (24 steps x r=7 frames)
    phases = [synthetic.gl_init_phase(900 + i, 1025, 24 * 7) for i in range(3)]
    one = syn.tts_ids(sents, init_phases=phases)
    bat = syn.tts_ids_batched(sents, init_phases=phases)
    sr1, a = scipy.io.wavfile.read(io.BytesIO(one.getvalue()))
    sr2, b = scipy.io.wavfile.read(io.BytesIO(bat.getvalue()))
    assert sr1 == sr2 == 22050 and a.dtype == b.dtype == np.int16
    assert a.shape == b.shape
    err = a.astype(np.float64) - b.astype(np.float64)
    snr = 10 * np.log10((a.astype(np.float64) ** 2).sum() / max((err ** 2).sum(), 1e-300))
    assert snr > 50.0, snr


def test_handles_release_their_device_memory():
    """Building, running and dropping both models several times leaves the free device memory where it was
    (within the allocator's slack): create / destroy pairs release the packed weights, workspaces and graphs."""
    import gc
    from tts_amd import synthetic
    from tts_amd.tacotron import Tacotron
    from tts_amd.tacotron2 import Tacotron2

    def cycle():
        m = Tacotron2(num_chars=129, num_speakers=0, r=7, attn_norm="sigmoid", prenet_type="bn", prenet_dropout=False,
                      separate_stopnet=True)
        m.load_state_dict(synthetic.t2_state_dict(seed=0, gains={"stop_bias": -30.0}))
        m.cuda().eval()
        m.inference_batch(torch.as_tensor(synthetic.t2_text_ids(3, 4, 30)), fixed_steps=3)
        v = Tacotron(61, 256, 1025, 80, 2, None, 5)
        v.load_state_dict(synthetic.t1_state_dict(seed=0))
        v.cuda().eval()
        v.decoder.max_decoder_steps = 3
        v.forward(torch.as_tensor(synthetic.t1_text_ids(1, 20)).cuda())
        del m, v
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    cycle()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(4):
        cycle()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, "leaked %.1f MB over four create/destroy cycles" % ((free0 - free1) / 2 ** 20)


def test_two_request_threads_with_their_own_models():
    """Two host threads, each with its own model handle and stream, sharing one AudioProcessor (whose vocoder
    calls serialise on the library's scratch lock): every result equals the one the same model gives alone."""
    import threading
    from tts_amd import synthetic
    from tts_amd.audio import AudioProcessor
    from tts_amd.tacotron2 import Tacotron2
    from oracle import audio_oracle as ao
    kw = {k: v for k, v in ao.T2_AUDIO.items() if k not in ("pad_mode", "win_from_hop")}
    kw["griffin_lim_iters"] = 5
    ap = AudioProcessor(verbose=False, **kw)
    models, ids = [], []
    for s in range(2):
        m = Tacotron2(num_chars=129, num_speakers=0, r=7, attn_norm="sigmoid", prenet_type="bn", prenet_dropout=False,
                      separate_stopnet=True)
        m.load_state_dict(synthetic.t2_state_dict(seed=s, gains={"stop_bias": -30.0}))
        m.cuda().eval()
        m.decoder.set_r(2)
        models.append(m)
        ids.append(torch.as_tensor(synthetic.t2_text_ids(40 + s, 1, 30 + 11 * s)))
    phase = [synthetic.gl_init_phase(60 + s, 1025, 24) for s in range(2)]

    def request(s):
        o = models[s].inference_batch(ids[s], fixed_steps=12)
        mel = o["postnet_output"][0].cpu().numpy().T
        return mel, ap.inv_mel_spectrogram(mel, init_phase=phase[s])

    alone = [request(s) for s in range(2)]
    got, errs = [[None] * 4 for _ in range(2)], []

    def worker(s):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for k in range(4):
                    got[s][k] = request(s)
        except Exception as e:   # surfaced below: an exception in a thread would otherwise pass silently
            errs.append(e)

    th = [threading.Thread(target=worker, args=(s,)) for s in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for s in range(2):
        for k in range(4):
            np.testing.assert_array_equal(got[s][k][0], alone[s][0])
            np.testing.assert_array_equal(got[s][k][1], alone[s][1])


def test_short_lived_request_threads_do_not_accumulate_vocoder_handles():
    """A threaded server starts a thread per request (Flask's app.run, Tacotron2/Tacotron2.py:59): threads that come and go,
    each with a stream of its own, one Griffin-Lim call each through one shared AudioProcessor.  The handles
    are pooled, so their number stays bounded and the free device memory stays where it was."""
    import threading
    from tts_amd import synthetic
    from tts_amd.audio import AudioProcessor
    from oracle import audio_oracle as ao
    kw = {k: v for k, v in ao.T2_AUDIO.items() if k not in ("pad_mode", "win_from_hop")}
    kw["griffin_lim_iters"] = 2
    ap = AudioProcessor(verbose=False, **kw)
    F = 400
    mel = synthetic.normal(5, "pool.mel", (80, F), 0.0, 1.0).astype(np.float32)
    ph = synthetic.gl_init_phase(6, 1025, F)
    want = ap.inv_mel_spectrogram(mel, init_phase=ph)
    errs = []

    def request(own_stream):
        try:
            if own_stream:
                with torch.cuda.stream(torch.cuda.Stream()):
                    got = ap.inv_mel_spectrogram(mel, init_phase=ph)
            else:
                got = ap.inv_mel_spectrogram(mel, init_phase=ph)
            np.testing.assert_array_equal(got, want)
        except Exception as e:
            errs.append(e)

    def burst(n, own_stream):
        for _ in range(n):
            t = threading.Thread(target=request, args=(own_stream,))
            t.start()
            t.join()
    burst(40, True)      # (past torch's pool of 32 streams per device: every stream the threads can get exists now)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()        # (torch caches blocks per stream; only the library's own allocations are of interest)
    free0 = torch.cuda.mem_get_info()[0]
    burst(20, True)
    burst(20, False)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert not errs, errs
    assert len(ap._handles) <= ap.MAX_IDLE_VOCODERS + 1, len(ap._handles)
    assert free0 - free1 < 8 << 20, "device memory grew by %.1f MB over forty request threads" % ((free0 - free1) / 2 ** 20)
