"""CPU oracle for the vocoder path (TEST INFRASTRUCTURE ONLY).

NumPy restatement of the inverse path of the reference `AudioProcessor`
(Tacotron2/TTS/utils.py:802-814, 834-878, 896-942; Tacotron/TTS/utils.py:
73-90, 109-151, 169-208).  Control flow, constants and the dtype trail follow
those lines.  Three calls in that code go to librosa 0.7.2
(scripts/create_venv.sh:8), which is neither vendored in the reference nor
installed here: `librosa.stft`, `librosa.istft` and `librosa.filters.mel`.
They are restated below from the library's published algorithm.

PARITY UNPINNED for those three calls: the reference holds no test, fixture or
golden vector for them and the library cannot be run.  What stands in:
`tests/test_audio_oracle.py` checks `stft`/`istft` here against
`torch.stft`/`torch.istft` (same conventions) and the mel basis against its
defining properties; the GPU path takes `inv_mel_basis` as an explicit input so
kernel parity does not depend on the mel-basis restatement.

Nothing in `tts_amd/` imports this file.
"""
import numpy as np
import scipy.signal


# ---------------------------------------------------------------- librosa 0.7.2 restatements
def hann_padded(win_length, n_fft):
    """`get_window('hann', win_length, fftbins=True)` zero-padded centrally to n_fft."""
    w = scipy.signal.get_window("hann", win_length, fftbins=True)
    lpad = (n_fft - win_length) // 2
    return np.pad(w, (lpad, n_fft - win_length - lpad), mode="constant")


def stft(y, n_fft, hop_length, win_length, pad_mode):
    """librosa.stft(center=True): float64 FFT, result stored as complex64."""
    w = hann_padded(win_length, n_fft)
    y = np.pad(y, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = y[idx]                                         # [n_fft, n_frames]
    return np.fft.rfft(w[:, None] * frames, axis=0).astype(np.complex64)


def window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=np.float32):
    x = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=dtype)
    win_sq = hann_padded(win_length, n_fft) ** 2
    for i in range(n_frames):
        s = i * hop_length
        x[s:min(len(x), s + n_fft)] += win_sq[:max(0, min(n_fft, len(x) - s))]
    return x


def istft(X, hop_length, win_length, dtype=np.float32):
    """librosa.istft(center=True, length=None): float64 inverse FFT, windowed,
    overlap-added frame by frame into a `dtype` buffer, divided by the window
    sum-square where it exceeds tiny, trimmed by n_fft//2 at both ends."""
    n_fft = 2 * (X.shape[0] - 1)
    w = hann_padded(win_length, n_fft)
    n_frames = X.shape[1]
    y = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=dtype)
    ytmp = w[:, None] * np.fft.irfft(X, n=n_fft, axis=0)
    for i in range(n_frames):
        s = i * hop_length
        y[s:s + n_fft] += ytmp[:, i]
    wss = window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=dtype)
    nz = wss > np.finfo(dtype).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = f >= min_log_hz
    mels = np.where(big, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    return np.where(big, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with the 0.7 defaults
    htk=False, norm=1 (Slaney): float32 [n_mels, 1 + n_fft // 2]."""
    if fmax is None:
        fmax = float(sr) / 2
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# ---------------------------------------------------------------- AudioProcessor inverse path
class AudioOracle:
    """Inverse path of `AudioProcessor`; constructor arguments are the `audio`
    block of the model config (Tacotron2/TTS/utils.py:747-789).  `pad_mode` is
    'constant' for the Tacotron2 tree (utils.py:937) and 'reflect' (librosa's
    default) for the Tacotron tree (Tacotron/TTS/utils.py:199-204);
    `win_from_hop` selects win_length = int(hop * factor) (Tacotron2,
    utils.py:853-860) or int(frame_length_ms / 1000 * sr) (Tacotron,
    Tacotron/TTS/utils.py:127-134)."""

    def __init__(self, sample_rate=22050, num_mels=80, min_level_db=-100, frame_shift_ms=12.5,
                 frame_length_ms=50, ref_level_db=20, num_freq=1025, power=1.5, preemphasis=0.98,
                 signal_norm=True, symmetric_norm=True, max_norm=4, mel_fmin=0.0, mel_fmax=8000.0,
                 clip_norm=True, griffin_lim_iters=60, pad_mode="constant", win_from_hop=True, **_):
        self.sample_rate, self.num_mels, self.min_level_db = sample_rate, num_mels, min_level_db or 0
        self.ref_level_db, self.num_freq, self.power = ref_level_db, num_freq, power
        self.preemphasis, self.signal_norm, self.symmetric_norm = preemphasis, signal_norm, symmetric_norm
        self.max_norm = 1.0 if max_norm is None else float(max_norm)
        self.mel_fmin, self.mel_fmax, self.clip_norm = mel_fmin or 0, mel_fmax, clip_norm
        self.griffin_lim_iters, self.pad_mode = griffin_lim_iters, pad_mode
        self.n_fft = (num_freq - 1) * 2
        self.hop_length = int(frame_shift_ms / 1000.0 * sample_rate)
        if win_from_hop:
            self.win_length = int(self.hop_length * (frame_length_ms / frame_shift_ms))
        else:
            self.win_length = int(frame_length_ms / 1000.0 * sample_rate)

    def _denormalize(self, S):
        if not self.signal_norm:
            return S
        if self.symmetric_norm:
            if self.clip_norm:
                S = np.clip(S, -self.max_norm, self.max_norm)
            return ((S + self.max_norm) * -self.min_level_db / (2 * self.max_norm)) + self.min_level_db
        if self.clip_norm:
            S = np.clip(S, 0, self.max_norm)
        return (S * -self.min_level_db / self.max_norm) + self.min_level_db

    @staticmethod
    def _db_to_amp(x):
        return np.power(10.0, x * 0.05)

    def _build_mel_basis(self):
        return mel_basis(self.sample_rate, self.n_fft, self.num_mels, self.mel_fmin, self.mel_fmax)

    def inv_mel_basis(self):
        return np.linalg.pinv(self._build_mel_basis())

    def _mel_to_linear(self, mel_spec, inv_basis=None):
        if inv_basis is None:
            inv_basis = self.inv_mel_basis()
        return np.maximum(1e-10, np.dot(inv_basis, mel_spec))

    def _stft(self, y):
        return stft(y, self.n_fft, self.hop_length, self.win_length, self.pad_mode)

    def _istft(self, Y):
        return istft(Y, self.hop_length, self.win_length)

    def _griffin_lim(self, S, init_phase01, return_all=False):
        """`init_phase01` in [0,1) replaces `np.random.rand(*S.shape)`."""
        angles = np.exp(2j * np.pi * init_phase01)
        S_complex = np.abs(S).astype(complex)
        y = self._istft(S_complex * angles)
        ys = [y]
        for _ in range(self.griffin_lim_iters):
            angles = np.exp(1j * np.angle(self._stft(y)))
            y = self._istft(S_complex * angles)
            if return_all:
                ys.append(y)
        return ys if return_all else y

    def apply_inv_preemphasis(self, x):
        if self.preemphasis == 0:
            raise RuntimeError(" [!] Preemphasis is set 0.0.")
        return scipy.signal.lfilter([1], [1, -self.preemphasis], x)

    def spec_from_mel(self, mel, inv_basis=None):
        D = self._denormalize(mel)
        S = self._db_to_amp(D + self.ref_level_db)
        S = self._mel_to_linear(S, inv_basis)
        return S ** self.power

    def spec_from_linear(self, spec):
        S = self._denormalize(spec)
        S = self._db_to_amp(S + self.ref_level_db)
        return S ** self.power

    def inv_mel_spectrogram(self, mel, init_phase01, inv_basis=None):
        y = self._griffin_lim(self.spec_from_mel(mel, inv_basis), init_phase01)
        return self.apply_inv_preemphasis(y) if self.preemphasis != 0 else y

    def inv_spectrogram(self, spec, init_phase01):
        y = self._griffin_lim(self.spec_from_linear(spec), init_phase01)
        return self.apply_inv_preemphasis(y) if self.preemphasis != 0 else y

    def find_endpoint(self, wav, threshold_db=-40, min_silence_sec=0.8):
        window_length = int(self.sample_rate * min_silence_sec)
        hop_length = int(window_length / 4)
        threshold = self._db_to_amp(threshold_db)
        for x in range(hop_length, len(wav) - window_length, hop_length):
            if np.max(wav[x:x + window_length]) < threshold:
                return x + hop_length
        return len(wav)


T2_AUDIO = dict(sample_rate=22050, num_mels=80, min_level_db=-100, frame_shift_ms=12.5,
                frame_length_ms=50, ref_level_db=20, num_freq=1025, power=1.5, preemphasis=0.98,
                signal_norm=True, symmetric_norm=True, max_norm=4, mel_fmin=0.0, mel_fmax=8000.0,
                clip_norm=True, griffin_lim_iters=60, pad_mode="constant", win_from_hop=True)
T1_AUDIO = dict(T2_AUDIO, symmetric_norm=False, max_norm=1, mel_fmin=None, mel_fmax=None,
                pad_mode="reflect", win_from_hop=False)
