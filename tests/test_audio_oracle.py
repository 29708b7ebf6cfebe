"""Audio oracle: pieces that the reference can run without librosa are pinned
by reference-generated fixtures; the librosa restatements (PARITY UNPINNED by
the reference) are cross-checked against torch.stft / torch.istft, which
follow the same conventions, and against defining properties."""
import json
import os

import numpy as np
import torch

from oracle import audio_oracle as ao

from conftest import GOLDEN


def test_pieces_match_reference_fixture():
    g = np.load(os.path.join(GOLDEN, "t2_audio_pieces.npz"))
    a = ao.AudioOracle(**ao.T2_AUDIO)
    assert [a.n_fft, a.hop_length, a.win_length] == list(g["params"])
    np.testing.assert_array_equal(a._denormalize(g["x"]), g["denorm"])
    np.testing.assert_array_equal(a._db_to_amp(a._denormalize(g["x"]) + a.ref_level_db), g["amp"])
    np.testing.assert_array_equal(a.apply_inv_preemphasis(g["y"]), g["deemph"])
    assert a.find_endpoint(g["wav"]) == int(g["endpoint"])


def test_stft_istft_against_torch():
    for cfg, pad in ((ao.T2_AUDIO, "constant"), (ao.T1_AUDIO, "reflect")):
        a = ao.AudioOracle(**cfg)
        rng = np.random.default_rng(0)
        y = rng.standard_normal(11000).astype(np.float32)
        w = torch.as_tensor(ao.hann_padded(a.win_length, a.n_fft))
        X = a._stft(y)
        Xt = torch.stft(torch.as_tensor(y, dtype=torch.float64), a.n_fft, a.hop_length, a.n_fft, window=w,
                        center=True, pad_mode=pad, return_complex=True).numpy()
        assert X.shape == Xt.shape == (1025, 41)
        assert np.abs(X - Xt).max() < 2e-5
        yi = a._istft(Xt)
        yt = torch.istft(torch.as_tensor(Xt), a.n_fft, a.hop_length, a.n_fft, window=w, center=True).numpy()
        assert yi.shape == yt.shape == (a.hop_length * 40,)
        assert np.abs(yi - yt).max() < 2e-6
        # a re-STFT of the ISTFT output has the same frame count: the GL loop is shape-stable
        assert a._stft(yi).shape == X.shape


def test_mel_basis_properties():
    mb = ao.mel_basis(22050, 2048, 80, 0.0, 8000.0)
    assert mb.shape == (80, 1025) and mb.dtype == np.float32
    assert (mb >= 0).all()
    freqs = np.linspace(0, 11025, 1025)
    assert mb[:, freqs > 8000].max() == 0          # nothing above fmax
    peaks = mb.argmax(1)
    assert (np.diff(peaks) > 0).all()              # centre frequencies increase
    # Slaney normalisation: each triangle integrates to ~1 in Hz
    area = mb.sum(1) * (11025 / 1024)
    assert np.abs(area - 1).max() < 0.08
    inv = np.linalg.pinv(mb)
    assert inv.shape == (1025, 80)
    assert np.abs(mb @ inv - np.eye(80)).max() < 1e-3


def test_griffin_lim_is_consistent():
    a = ao.AudioOracle(**dict(ao.T2_AUDIO, griffin_lim_iters=8))
    rng = np.random.default_rng(1)
    y = (rng.standard_normal(275 * 29) * 0.1).astype(np.float32)
    S = np.abs(a._stft(y)).astype(np.float32)
    ph = rng.random(S.shape)
    ys = a._griffin_lim(S, ph, return_all=True)
    assert ys[-1].shape == (275 * 29,)
    # spectral convergence improves over the iterations
    err = [np.linalg.norm(np.abs(a._stft(v)) - S) / np.linalg.norm(S) for v in (ys[0], ys[-1])]
    assert err[1] < err[0]


def test_package_host_helpers_match_reference_fixture():
    """The NumPy-in / NumPy-out helpers AudioProcessor keeps under the reference's names, against values the
    reference itself produced (no GPU involved)."""
    import io
    from tts_amd.audio import AudioProcessor
    for name, v1 in (("t2_audio_pieces.npz", False), ("t1_audio_pieces.npz", True)):
        g = np.load(os.path.join(GOLDEN, name))
        meta = json.loads(str(g["meta"]))
        ap = AudioProcessor(verbose=False, tacotron_v1=v1, **meta)
        assert [ap.n_fft, ap.hop_length, ap.win_length] == g["params"].tolist()
        den = ap._denormalize(g["x"])
        np.testing.assert_array_equal(den, g["denorm"])
        np.testing.assert_array_equal(ap._db_to_amp(den + ap.ref_level_db), g["amp"])
        if v1:
            continue
        np.testing.assert_array_equal(ap.apply_inv_preemphasis(g["y"]), g["deemph"])
        assert ap.find_endpoint(g["wav"]) == int(g["endpoint"])
        assert ap.find_endpoint(g["wav"][:100]) == 100
        buf = io.BytesIO()
        ap.save_wav(g["deemph"], buf)
        assert np.array_equal(np.frombuffer(buf.getvalue(), dtype=np.uint8), g["wav_bytes"])


def test_mel_filterbank_has_the_published_slaney_properties():
    """PARITY UNPINNED: librosa.filters.mel is not importable here and the reference holds no vectors for it.
    What can be checked without it are the properties the published Slaney construction defines, on the
    oracle's filterbank and on the package's (two separately written functions; the last assertion only says
    that they agree with each other, which pins neither)."""
    from tts_amd import audio
    # mel scale: 200/3 Hz per mel below 1 kHz, 27 mels per factor 6.4 above
    assert abs(float(audio._hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(audio._hz_to_mel(6400.0)) - 42.0) < 1e-9
    assert abs(float(audio._mel_to_hz(42.0)) - 6400.0) < 1e-6 and abs(float(audio._mel_to_hz(7.5)) - 500.0) < 1e-9
    orc = ao.AudioOracle(**ao.T2_AUDIO)
    for fb in (orc._build_mel_basis(), audio.mel_filterbank(22050, 2048, 80, 0.0, 8000.0)):
        assert fb.shape == (80, 1025) and fb.dtype == np.float32 and (fb >= 0).all()
        freqs = np.linspace(0, 11025.0, 1025)
        assert fb[:, freqs > 8000.0].max() == 0.0                       # nothing above fmax
        peaks = fb.argmax(axis=1)
        assert (np.diff(peaks) > 0).all()                               # centres increase
        for row in fb:                                                  # each filter is one rise and one fall
            nz = np.nonzero(row)[0]
            assert (np.diff(nz) == 1).all()
            k = row.argmax()
            assert (np.diff(row[nz[0]:k + 1]) >= 0).all() and (np.diff(row[k:nz[-1] + 1]) <= 0).all()
        # Slaney normalisation: every triangle has unit area in Hz (up to the sampling of the triangle)
        area = fb.sum(axis=1) * (11025.0 / 1024)
        assert np.abs(area - 1.0).max() < 0.15 and abs(area[40:].mean() - 1.0) < 0.01
        # centre of filter i is mel point i+1 of 82 equally spaced mel points between 0 and 8000 Hz
        mel_pts = audio._mel_to_hz(np.linspace(audio._hz_to_mel(0.0), audio._hz_to_mel(8000.0), 82))
        assert np.abs(freqs[peaks] - mel_pts[1:-1]).max() <= 11025.0 / 1024
    np.testing.assert_array_equal(orc._build_mel_basis(), audio.mel_filterbank(22050, 2048, 80, 0.0, 8000.0))


def _gl_cases():
    for name, cfg, use_mel in (("t2gl_reference_loop.npz", ao.T2_AUDIO, True), ("t1gl_reference_loop.npz", ao.T1_AUDIO, False)):
        g = np.load(os.path.join(GOLDEN, name))
        meta = json.loads(str(g["meta"]))
        for F, seed in zip(meta["frames"], meta["seeds"]):
            yield name, cfg, use_mel, g, F, seed


def test_oracle_replays_the_references_own_griffin_lim_loop():
    """`t2gl` / `t1gl` fixtures: the reference's `_griffin_lim`, `inv_mel_spectrogram` (Tacotron2/TTS/utils.py:
    905-929) and `inv_spectrogram` (Tacotron/TTS/utils.py:169-196) themselves, run with np.random.seed(k) and a
    placeholder librosa whose stft / istft / filters.mel are this oracle's restatements.  Pins the oracle's loop,
    dtype trail (complex128 product, float32 signal, complex64 spectrum), angle(0) = 0 (frames 3..9 of S are
    zero), pinv, power and de-emphasis against the reference's code; librosa itself stays PARITY UNPINNED."""
    for name, cfg, use_mel, g, F, seed in _gl_cases():
        a = ao.AudioOracle(**cfg)
        assert [a.n_fft, a.hop_length, a.win_length] == g["params"].tolist()
        S, y_ref = g["S%d" % F], g["y%d" % F]
        assert (S[:, 3:10] == 0).all() and y_ref.dtype == np.float32
        y = a._griffin_lim(S, np.random.RandomState(seed).rand(1025, F))
        assert y.dtype == np.float32 and y.shape == y_ref.shape == (a.hop_length * (F - 1),)
        assert np.abs(y - y_ref).max() <= 1e-6 * max(1.0, np.abs(y_ref).max()), (name, F)
        ph = np.random.RandomState(seed + 1).rand(1025, F)
        wav = a.inv_mel_spectrogram(g["x%d" % F], ph) if use_mel else a.inv_spectrogram(g["x%d" % F], ph)
        wav_ref = g["wav%d" % F]
        assert wav.dtype == wav_ref.dtype == np.float64
        assert np.abs(wav - wav_ref).max() <= 1e-6 * np.abs(wav_ref).max(), (name, F)
