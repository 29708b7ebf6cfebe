"""GPU parity of the vocoder path (spec -> magnitudes, STFT, ISTFT, Griffin-Lim, de-emphasis)
against the CPU oracle.  Bars: STFT/ISTFT max-abs 1e-4 relative to signal scale, Griffin-Lim
waveform SNR >= 40 dB against the oracle started from the same initial phase."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _snr_db(ref, got):
    ref = np.asarray(ref, dtype=np.float64)
    err = ref - np.asarray(got, dtype=np.float64)
    return 10 * np.log10((ref ** 2).sum() / max((err ** 2).sum(), 1e-300))


def _aps(v1=False, iters=60):
    from oracle import audio_oracle as ao
    from tts_amd.audio import AudioProcessor
    cfg = dict(ao.T1_AUDIO if v1 else ao.T2_AUDIO, griffin_lim_iters=iters)
    orc = ao.AudioOracle(**cfg)
    kw = {k: v for k, v in cfg.items() if k not in ("pad_mode", "win_from_hop")}
    ap = AudioProcessor(tacotron_v1=v1, verbose=False, **kw)
    return orc, ap


def _smooth_mel(seed, n_mels, F, scale=1.5):
    from tts_amd import synthetic
    m = synthetic.normal(seed, "mel", (n_mels, F + 4), 0.0, scale)
    return np.stack([np.convolve(r, np.ones(5) / 5, mode="valid") for r in m]).astype(np.float32)


@pytest.mark.parametrize("v1", [False, True])
def test_stft_istft_match_oracle(v1):
    orc, ap = _aps(v1)
    assert (ap.n_fft, ap.hop_length, ap.win_length) == (orc.n_fft, orc.hop_length, orc.win_length)
    rng = np.random.default_rng(3)
    for F in (2, 5, 9, 64):
        y = (rng.standard_normal(ap.hop_length * (F - 1)) * 0.3).astype(np.float32)
        X_ref = orc._stft(y)
        X = ap._stft(y)
        assert X.shape == X_ref.shape == (1025, F)
        assert np.abs(X - X_ref).max() < 1e-4 * max(1.0, np.abs(X_ref).max())
        y_ref = orc._istft(X_ref.astype(np.complex128))
        y_got = ap._istft(X_ref)
        assert y_got.shape == y_ref.shape
        assert np.abs(y_got - y_ref).max() < 1e-5


def test_spec_to_mag_matches_oracle():
    orc, ap = _aps(False)
    mel = _smooth_mel(11, 80, 57, 2.5)
    ref = orc.spec_from_mel(mel, ap._build_mel_basis().__class__ and np.linalg.pinv(ap._build_mel_basis()))
    got = ap.spec_to_mag(torch.from_numpy(mel.T.copy()).cuda().unsqueeze(0))[0].cpu().numpy().T
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)
    # fp32 sums of 80 terms with cancellation: compare against the magnitude scale of each frame
    assert (np.abs(got - ref) / (np.abs(ref).max(axis=0, keepdims=True) + 1e-12)).max() < 1e-4, rel.max()
    orc1, ap1 = _aps(True)
    lin = np.clip(_smooth_mel(12, 1025, 9, 0.4) + 0.5, -0.2, 1.2)
    ref = orc1.spec_from_linear(lin)
    got = ap1.spec_to_mag(torch.from_numpy(lin.T.copy()).cuda().unsqueeze(0), mel=False)[0].cpu().numpy().T
    assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1e-9)).max() < 1e-4


@pytest.mark.parametrize("v1,F", [(False, 40), (False, 173), (True, 61)])
def test_griffin_lim_snr(v1, F):
    from tts_amd import synthetic
    orc, ap = _aps(v1, iters=60)
    if v1:
        spec = np.clip(_smooth_mel(21, 1025, F, 0.25) + 0.45, 0.0, 1.0)
        S = orc.spec_from_linear(spec)
    else:
        mel = _smooth_mel(22, 80, F, 1.5)
        S = orc.spec_from_mel(mel)
    S = S.astype(np.float32)
    ph = synthetic.gl_init_phase(9, 1025, F)
    y_ref = orc._griffin_lim(S, ph)
    y = ap._griffin_lim(S, init_phase=ph)
    assert y.shape == y_ref.shape == (ap.hop_length * (F - 1),)
    snr = _snr_db(y_ref, y)
    assert snr >= 40.0, "Griffin-Lim SNR %.1f dB" % snr


def test_inv_mel_spectrogram_end_to_end_and_ragged_batch():
    from tts_amd import synthetic
    orc, ap = _aps(False, iters=30)
    Fs = [33, 50, 12]
    mels = [_smooth_mel(40 + i, 80, F, 1.5) for i, F in enumerate(Fs)]
    phs = [synthetic.gl_init_phase(50 + i, 1025, F) for i, F in enumerate(Fs)]
    refs = [orc.inv_mel_spectrogram(m, p, np.linalg.pinv(ap._build_mel_basis())) for m, p in zip(mels, phs)]
    # reference-shaped call, one utterance
    y0 = ap.inv_mel_spectrogram(mels[0], init_phase=phs[0])
    assert y0.dtype == np.float64 and y0.shape == refs[0].shape
    assert _snr_db(refs[0], y0) >= 40.0
    # ragged batch through the batched device API
    Fm = max(Fs)
    spec = torch.zeros(3, Fm, 80)
    ph = torch.zeros(3, Fm, 1025)
    for b, F in enumerate(Fs):
        spec[b, :F] = torch.from_numpy(mels[b].T.copy())
        ph[b, :F] = torch.from_numpy(phs[b].T.astype(np.float32).copy())
    mag = ap.spec_to_mag(spec.cuda())
    wav = ap.griffin_lim_batch(mag, ph.cuda(), torch.tensor(Fs)).cpu().numpy()
    for b, F in enumerate(Fs):
        L = ap.hop_length * (F - 1)
        assert _snr_db(refs[b], wav[b, :L]) >= 40.0, b
        assert np.abs(wav[b, L:]).max(initial=0.0) == 0.0
    # bit-reproducible: no atomics in the overlap-add
    wav2 = ap.griffin_lim_batch(mag, ph.cuda(), torch.tensor(Fs)).cpu().numpy()
    np.testing.assert_array_equal(wav, wav2)


def test_vocoder_argument_errors():
    from tts_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 4, 1025, device="cuda")
    w = torch.zeros(1, 275 * 3, device="cuda")
    bad = _lib.StftParams(1024, 275, 1100, 0)
    rc = lib.ttsx_griffin_lim(_lib.ptr(x), _lib.ptr(x), None, 1, 4, bad, 1, 0.98, _lib.ptr(w), _lib.current_stream())
    assert rc == _lib.ERR_ARG and b"n_fft" in lib.ttsx_last_error_string()


def test_griffin_lim_two_frames_and_zero_iterations():
    """The shortest legal utterance (two frames = one hop of signal) and iters = 0 (one ISTFT of
    the initial phase)."""
    from tts_amd import synthetic
    for iters in (0, 3):
        orc, ap = _aps(False, iters=iters)
        S = np.abs(_smooth_mel(3, 1025, 2, 1.0)).astype(np.float32)
        ph = synthetic.gl_init_phase(4, 1025, 2)
        y_ref = orc._griffin_lim(S, ph)
        y = ap._griffin_lim(S, init_phase=ph)
        assert y.shape == y_ref.shape == (ap.hop_length,)
        assert _snr_db(y_ref, y) >= 40.0


def test_griffin_lim_is_scale_invariant_over_the_float_range():
    """Griffin-Lim is positively homogeneous: magnitudes scaled by a power of two give the same waveform scaled
    by it.  At 2^-70 and 2^+52 the squared bin magnitudes leave the float32 range, so the phase normalisation
    takes its rescaling branch; silence (all-zero magnitudes) must give exact zeros, not NaN."""
    from tts_amd import synthetic
    orc, ap = _aps(False, iters=12)
    F = 24
    S = orc.spec_from_mel(_smooth_mel(31, 80, F, 1.5)).astype(np.float32)
    ph = synthetic.gl_init_phase(32, 1025, F)
    y = ap._griffin_lim(S, init_phase=ph)
    for e in (-70, 52):
        c = np.float32(2.0) ** e
        ys = ap._griffin_lim(S * c, init_phase=ph)
        assert np.isfinite(ys).all(), e
        assert _snr_db(y, ys.astype(np.float64) / float(c)) >= 60.0, e
    z = ap._griffin_lim(np.zeros_like(S), init_phase=ph)
    assert np.array_equal(z, np.zeros_like(z))


@pytest.mark.parametrize("hop,win,v1", [(137, 1096, False), (1100, 1100, False), (64, 512, False), (551, 1101, True),
                                        (300, 1102, True), (1, 2, False)])
def test_stft_geometries_at_the_limits(hop, win, v1):
    """Window / hop combinations at the edges of what the entry points accept: eight overlapping frames per
    sample, no overlap at all (hop = win: no carry between rounds), a short window, an odd window with odd left
    padding, the longest window (1102), the shortest (2); STFT, ISTFT and a short Griffin-Lim run against the
    oracle.  (sample_rate = 1000 makes milliseconds equal samples.)"""
    from oracle import audio_oracle as ao
    from tts_amd import synthetic
    from tts_amd.audio import AudioProcessor
    cfg = dict(ao.T1_AUDIO if v1 else ao.T2_AUDIO, sample_rate=1000, frame_shift_ms=float(hop),
               frame_length_ms=float(win), mel_fmin=0.0, mel_fmax=None, griffin_lim_iters=4)
    orc = ao.AudioOracle(**cfg)
    ap = AudioProcessor(tacotron_v1=v1, verbose=False,
                        **{k: v for k, v in cfg.items() if k not in ("pad_mode", "win_from_hop")})
    assert (ap.hop_length, ap.win_length) == (orc.hop_length, orc.win_length) == (hop, win)
    F = max(9, 1100 // hop + 6) if hop > 1 else 1500        # (reflect padding needs more than n_fft/2 samples)
    rng = np.random.default_rng(hop * 7 + win)
    y = (rng.standard_normal(hop * (F - 1)) * 0.3).astype(np.float32)
    X_ref = orc._stft(y)
    X = ap._stft(y)
    assert X.shape == X_ref.shape == (1025, F)
    assert np.abs(X - X_ref).max() < 1e-4 * max(1.0, np.abs(X_ref).max())
    y_ref = orc._istft(X_ref.astype(np.complex128))
    y_got = ap._istft(X_ref)
    assert y_got.shape == y_ref.shape
    d = np.abs(y_got - y_ref) / max(1.0, np.abs(y_ref).max())
    if hop == win:
        # no overlap: next to the frame boundaries the signal is (x w^2) / w^2 with w^2 down to 1e-11, where fp32
        # and the oracle's fp64 part ways by construction; everywhere else the usual bar holds
        assert np.quantile(d, 0.75) < 1e-5 and d.max() < 5e-2   # (w^2 < 1e-2 on a fifth of each frame)
    else:
        assert d.max() < 1e-5
    S = np.abs(X_ref).astype(np.float32)
    ph = synthetic.gl_init_phase(11, 1025, F)
    g_ref = orc._griffin_lim(S, ph)
    g = ap._griffin_lim(S, init_phase=ph)
    assert g.shape == g_ref.shape
    assert _snr_db(g_ref, g) >= 40.0


@pytest.mark.parametrize("hop,win", [(275, 1100), (256, 1024), (270, 1080), (274, 1096)])
def test_griffin_lim_long_runs_equal_one_utterance_runs(hop, win):
    """A ragged batch whose workgroups own several rounds of four frames each (the overlap-add of a round in the
    interior of an utterance takes the pass-per-hop-block form where the window is four hops of 256..292 samples;
    (274, 1096) is four hops too, but with an even left padding of the window the other three do not have)
    against the same utterances run one at a time (one round per workgroup, carry from the neighbour, general
    overlap-add): the same bits; and against the oracle."""
    from oracle import audio_oracle as ao
    from tts_amd import synthetic
    from tts_amd.audio import AudioProcessor
    cfg = dict(ao.T2_AUDIO, sample_rate=1000, frame_shift_ms=float(hop), frame_length_ms=float(win), mel_fmin=0.0,
               mel_fmax=None, griffin_lim_iters=3)
    orc = ao.AudioOracle(**cfg)
    ap = AudioProcessor(verbose=False, **{k: v for k, v in cfg.items() if k not in ("pad_mode", "win_from_hop")})
    assert (ap.hop_length, ap.win_length) == (hop, win)
    Fs = [700, 699, 698, 697, 333, 9]
    B, Fm = len(Fs), max(Fs)
    gen = torch.Generator().manual_seed(hop)
    mag = torch.rand(B, Fm, 1025, generator=gen) + 0.05
    ph = torch.rand(B, Fm, 1025, generator=gen)
    wav = ap.griffin_lim_batch(mag.cuda(), ph.cuda(), torch.tensor(Fs), deemphasis=False).cpu().numpy()
    for b, F in enumerate(Fs):
        L = hop * (F - 1)
        one = ap.griffin_lim_batch(mag[b:b + 1, :F].contiguous().cuda(), ph[b:b + 1, :F].contiguous().cuda(),
                                   torch.tensor([F]), deemphasis=False).cpu().numpy()
        assert np.array_equal(one[0, :L], wav[b, :L]), b
        assert np.abs(wav[b, L:]).max(initial=0.0) == 0.0
    for b in (1, 5):
        F = Fs[b]
        ref = orc._griffin_lim(mag[b, :F].numpy().T.copy(), ph[b, :F].numpy().T.astype(np.float64).copy())
        assert _snr_db(ref, wav[b, :hop * (F - 1)]) >= 40.0, b


def test_device_de_emphasis_equals_lfilter():
    """The blocked-scan de-emphasis kernel against scipy.signal.lfilter (fp64) applied to the same raw Griffin-Lim
    output, over several workgroup chunks (16 384 samples each) and on a ragged batch; and the coefficient range
    it accepts."""
    import scipy.signal
    from tts_amd import _lib
    orc, ap = _aps(False, iters=2)
    Fs = [330, 61, 200]
    rng = np.random.default_rng(5)
    mag = torch.from_numpy(rng.random((3, max(Fs), 1025), dtype=np.float32) * 2).cuda()
    ph = torch.from_numpy(rng.random((3, max(Fs), 1025), dtype=np.float32)).cuda()
    nf = torch.tensor(Fs)
    saved = ap.preemphasis
    try:
        ap.preemphasis = 0.0
        raw = ap.griffin_lim_batch(mag, ph, nf).cpu().numpy()
        ap.preemphasis = saved
        de = ap.griffin_lim_batch(mag, ph, nf).cpu().numpy()
        for b, F in enumerate(Fs):
            L = ap.hop_length * (F - 1)
            ref = scipy.signal.lfilter([1], [1, -saved], raw[b, :L].astype(np.float64))
            assert np.abs(de[b, :L] - ref).max() <= 1e-6 * np.abs(ref).max(), b
            assert np.abs(de[b, L:]).max(initial=0.0) == 0.0
        ap.preemphasis = 0.995      # |a|^2048 = 3.5e-5: beyond what the truncated scan represents
        with pytest.raises(ValueError):
            ap.griffin_lim_batch(mag, ph, nf)
    finally:
        ap.preemphasis = saved


def test_out_of_range_lengths_are_clamped_on_the_device():
    """n_frames entries above F or below 0 behave as F and 0: nothing outside the call's buffers is touched."""
    from tts_amd import _lib
    orc, ap = _aps(False, iters=6)
    F = 37
    gen = torch.Generator(device="cuda").manual_seed(5)
    mag = torch.rand(3, F, 1025, device="cuda", generator=gen) + 0.1
    ph = torch.rand(3, F, 1025, device="cuda", generator=gen)
    guard = torch.full((3, 275 * (F - 1) + 512), 7.0, device="cuda")
    good = ap.griffin_lim_batch(mag, ph, torch.tensor([F, 0, 20]))
    bad = ap.griffin_lim_batch(mag, ph, torch.tensor([F + 500, -9, 20]))
    assert torch.equal(good, bad) and torch.isfinite(bad).all()
    assert float(good[1].abs().max()) == 0.0 and float(good[2, 275 * 19:].abs().max()) == 0.0
    assert float(guard.min()) == 7.0


def test_two_vocoder_handles_on_two_streams_give_the_same_samples():
    """Handles share nothing: the same call through two handles, enqueued on two streams without any
    synchronisation in between, gives bit-identical waveforms (and the one-shot form agrees with both)."""
    import ctypes
    from tts_amd import _lib
    lib = _lib.load()
    sp = _lib.StftParams(2048, 275, 1100, _lib.PAD_CONSTANT)
    F, B = 90, 2
    gen = torch.Generator(device="cuda").manual_seed(8)
    mag = torch.rand(B, F, 1025, device="cuda", generator=gen) + 0.05
    ph = torch.rand(B, F, 1025, device="cuda", generator=gen)
    outs, handles, streams = [], [], [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for s in streams:
        h = ctypes.c_void_p()
        _lib.check(lib.ttsx_vocoder_create(ctypes.byref(sp), ctypes.c_void_p(s.cuda_stream), ctypes.byref(h)), "create")
        handles.append(h)
        outs.append(torch.empty(B, 275 * (F - 1), device="cuda"))
    for rep in range(2):     # the second round reuses each handle's scratch
        for h, s, w in zip(handles, streams, outs):
            _lib.check(lib.ttsx_vocoder_griffin_lim(h, _lib.ptr(mag), _lib.ptr(ph), None, B, F, 12, 0.98, _lib.ptr(w),
                                                    ctypes.c_void_p(s.cuda_stream)), "griffin_lim")
    torch.cuda.synchronize()
    one = torch.empty_like(outs[0])
    _lib.check(lib.ttsx_griffin_lim(_lib.ptr(mag), _lib.ptr(ph), None, B, F, ctypes.byref(sp), 12, 0.98, _lib.ptr(one),
                                    _lib.current_stream()), "one-shot")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], one)
    assert float(one.abs().max()) > 0
    for h in handles:
        assert lib.ttsx_vocoder_destroy(h) == _lib.OK


@pytest.mark.parametrize("name,v1", [("t2gl_reference_loop.npz", False), ("t1gl_reference_loop.npz", True)])
def test_griffin_lim_against_the_references_own_loop(name, v1):
    """Fixtures produced by the reference's `_griffin_lim` / `inv_mel_spectrogram` / `inv_spectrogram` themselves
    (seeded global NumPy RNG; the three librosa calls inside them were the oracle's restatements, so this pins the
    loop around them - 60 iterations, angle(0) = 0 on the silent frames 3..9, pinv, power, de-emphasis - and not
    librosa).  Waveform SNR >= 40 dB from the same initial phase."""
    import json
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(str(g["meta"]))
    _, ap = _aps(v1, iters=60)
    assert [ap.n_fft, ap.hop_length, ap.win_length] == g["params"].tolist()
    for F, seed in zip(meta["frames"], meta["seeds"]):
        y = ap._griffin_lim(g["S%d" % F], init_phase=np.random.RandomState(seed).rand(1025, F))
        assert y.shape == g["y%d" % F].shape
        snr = _snr_db(g["y%d" % F], y)
        assert snr >= 40.0, "%s F=%d _griffin_lim SNR %.1f dB" % (name, F, snr)
        ph = np.random.RandomState(seed + 1).rand(1025, F)
        wav = ap.inv_spectrogram(g["x%d" % F], init_phase=ph) if v1 else ap.inv_mel_spectrogram(g["x%d" % F], init_phase=ph)
        assert wav.dtype == np.float64 and wav.shape == g["wav%d" % F].shape
        snr = _snr_db(g["wav%d" % F], wav)
        assert snr >= 40.0, "%s F=%d end-to-end SNR %.1f dB" % (name, F, snr)


@pytest.mark.parametrize("B,F,iters", [(1, 402, 60), (1, 6, 5), (3, 61, 8), (48, 300, 4)])
def test_reflect_pad_written_by_the_launches_equals_the_pad_kernel(B, F, iters):
    """Tacotron (v1) geometry ('reflect' centre padding).  With utterances of full length the Griffin-Lim launches
    write the pad images themselves; with a length array (here: every length = F) a separate kernel fills the pad
    between the launches.  Same samples either way, from one workgroup per round (B = 1) to runs of several rounds
    with halo rounds (48 x 300); F = 6 is the shortest signal with one image per pad position + margin."""
    _, ap = _aps(True, iters=iters)
    g = torch.Generator(device="cpu").manual_seed(5 + F)
    mag = (torch.rand(B, F, 1025, generator=g) * 2).cuda()
    ph = torch.rand(B, F, 1025, generator=g).cuda()
    own = ap.griffin_lim_batch(mag, ph)
    by_kernel = ap.griffin_lim_batch(mag, ph, n_frames=torch.full((B,), F, dtype=torch.int32))
    assert own.shape == by_kernel.shape
    assert torch.isfinite(own).all()
    assert torch.equal(own, by_kernel)
