"""AudioProcessor with the reference's surface, inverse path on the GPU.

Mirrors the serving-time part of `AudioProcessor`
(Tacotron2/TTS/utils.py:746-951; Tacotron/TTS/utils.py:20-217): constructor
arguments and attributes, `inv_mel_spectrogram`, `inv_spectrogram`,
`_griffin_lim`, `_stft`, `_istft`, `_denormalize`, `_db_to_amp`,
`_mel_to_linear`, `_build_mel_basis`, `apply_inv_preemphasis`, `find_endpoint`,
`save_wav`.  The analysis path (`spectrogram`, `melspectrogram`, `load_wav`,
...) is training-time code and is not provided.

Differences, all additive: `_griffin_lim`, `inv_mel_spectrogram` and
`inv_spectrogram` accept `init_phase=` (values in [0,1), shape of S) so that a
run can be reproduced; when omitted the phase comes from `np.random.rand`, as
in the reference (utils.py:923).  `tacotron_v1=True` selects the Tacotron
tree's STFT parameters (win_length from frame_length_ms, reflect padding).
"""
import contextlib
import ctypes
import io
import threading
import weakref

import numpy as np
import scipy.io.wavfile
import torch

from . import _lib


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    out = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, out)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Slaney-style mel filterbank (what `librosa.filters.mel` of librosa 0.7
    returns with htk=False, norm=1): float32 [n_mels, 1 + n_fft//2]."""
    fmax = float(sr) / 2 if fmax is None else fmax
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        weights[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights


class AudioProcessor(object):
    def __init__(self, sample_rate=None, num_mels=None, min_level_db=None, frame_shift_ms=None,
                 frame_length_ms=None, ref_level_db=None, num_freq=None, power=None, preemphasis=0.0,
                 signal_norm=None, symmetric_norm=None, max_norm=None, mel_fmin=None, mel_fmax=None,
                 clip_norm=True, griffin_lim_iters=None, do_trim_silence=False, sound_norm=False,
                 tacotron_v1=False, verbose=True, device=None, **_):
        if verbose:
            print("\t[-] Setting up Audio Processor...")
        self.sample_rate = sample_rate
        self.num_mels = num_mels
        self.min_level_db = min_level_db or 0
        self.frame_shift_ms = frame_shift_ms
        self.frame_length_ms = frame_length_ms
        self.ref_level_db = ref_level_db
        self.num_freq = num_freq
        self.power = power
        self.preemphasis = preemphasis
        self.griffin_lim_iters = griffin_lim_iters
        self.signal_norm = signal_norm
        self.symmetric_norm = symmetric_norm
        self.mel_fmin = mel_fmin or 0
        self.mel_fmax = mel_fmax
        self.max_norm = 1.0 if max_norm is None else float(max_norm)
        self.clip_norm = clip_norm
        self.do_trim_silence = do_trim_silence
        self.sound_norm = sound_norm
        self.tacotron_v1 = tacotron_v1
        self.n_fft, self.hop_length, self.win_length = self._stft_parameters()
        assert min_level_db != 0.0, " [!] min_level_db is 0"
        if verbose:
            for key, value in vars(self).items():
                print("\t | > {}:{}".format(key, value))
        self._device = device
        self._inv_basis_dev = None
        # vocoder handles (a handle serves one caller at a time): a pool, not one per thread, so that a server
        # whose request threads come and go keeps as many handles as it ever had calls in flight at once
        self._handles = []                  # every handle this object has created (destroyed with it)
        self._idle = {}                     # stream -> handles not in use whose last work went to that stream
        self._handles_mu = threading.Lock()
        weakref.finalize(self, _destroy_vocoders, self._handles, self._handles_mu)

    # ---- host-side helpers, same arithmetic as the reference ---------------
    def _stft_parameters(self):
        n_fft = (self.num_freq - 1) * 2
        hop_length = int(self.frame_shift_ms / 1000.0 * self.sample_rate)
        if self.tacotron_v1:
            win_length = int(self.frame_length_ms / 1000.0 * self.sample_rate)
        else:
            factor = self.frame_length_ms / self.frame_shift_ms
            assert (factor).is_integer(), " [!] frame_shift_ms should divide frame_length_ms"
            win_length = int(hop_length * factor)
        return n_fft, hop_length, win_length

    # ---- host-side mirrors of small reference helpers ---------------------------------------------------
    # The serving path does this arithmetic on the device (ttsx_spec_to_mag: denormalise, dB -> amplitude,
    # mel -> linear; ttsx_request_tail: endpoint search, int16 scaling; the de-emphasis kernel).  The methods
    # below keep the reference's NumPy-in / NumPy-out helpers available under their names for callers that
    # use them directly; tests pin them to values the reference itself produced.
    def save_wav(self, wav, path):
        """utils.py:794-796: scale so that the peak (at least 0.01) maps to 32767, truncate to int16."""
        wav = np.asarray(wav)
        peak = np.abs(wav).max()
        gain = 32767 / max(0.01, peak)
        scipy.io.wavfile.write(path, self.sample_rate, (wav * gain).astype(np.int16))

    def _build_mel_basis(self):
        if self.mel_fmax is not None:
            assert self.mel_fmax <= self.sample_rate // 2
        return mel_filterbank(self.sample_rate, self.n_fft, self.num_mels, self.mel_fmin, self.mel_fmax)

    def _mel_to_linear(self, mel_spec):
        """utils.py:802-804 (the reference recomputes the pseudo-inverse on every call; so does this mirror -
        the device path keeps it resident, `inv_mel_basis_device`)."""
        return np.maximum(1e-10, np.linalg.pinv(self._build_mel_basis()) @ mel_spec)

    def _denormalize(self, S):
        """utils.py:834-851: normalised range [-max_norm, max_norm] (symmetric) or [0, max_norm] back to
        [min_level_db, 0] dB, clipping first when clip_norm is set."""
        if not self.signal_norm:
            return S
        lo, span = (-self.max_norm, 2 * self.max_norm) if self.symmetric_norm else (0, self.max_norm)
        if self.clip_norm:
            S = np.clip(S, lo, self.max_norm)
        if self.symmetric_norm:
            S = S + self.max_norm
        return (S * -self.min_level_db / span) + self.min_level_db

    @staticmethod
    def _db_to_amp(x):
        return np.power(10.0, x * 0.05)

    def apply_inv_preemphasis(self, x):
        if self.preemphasis == 0:
            raise RuntimeError(" [!] Preemphasis is set 0.0.")
        import scipy.signal
        return scipy.signal.lfilter([1], [1, -self.preemphasis], x)

    def find_endpoint(self, wav, threshold_db=-40, min_silence_sec=0.8):
        """utils.py:944-951: end of the first window of `min_silence_sec` (stepped by a quarter of it, starting
        one step in) whose maximum stays below the threshold; len(wav) if there is none."""
        wav = np.asarray(wav)
        window = int(self.sample_rate * min_silence_sec)
        step = int(window / 4)
        threshold = self._db_to_amp(threshold_db)
        starts = np.arange(step, len(wav) - window, step)
        if len(starts) == 0:
            return len(wav)
        for x in starts:
            if wav[x:x + window].max() < threshold:
                return int(x + step)
        return len(wav)

    # ---- device plumbing ----------------------------------------------------
    def _dev(self):
        if self._device is None:
            if not torch.cuda.is_available():
                raise _lib.TtsxError("AudioProcessor's inverse path runs on the GPU only (no CPU path)")
            self._device = torch.device("cuda", torch.cuda.current_device())
        return self._device

    def _stft_params(self):
        return _lib.StftParams(self.n_fft, self.hop_length, self.win_length,
                               _lib.PAD_REFLECT if self.tacotron_v1 else _lib.PAD_CONSTANT)

    MAX_IDLE_VOCODERS = 8

    @contextlib.contextmanager
    def _vocoder(self):
        """A vocoder handle (tables + scratch on this AudioProcessor's device) for the duration of one call,
        taken from the pool of idle handles or created.  Calls are stream-ordered and return before the GPU has
        finished, so an idle handle is only handed to a caller on the stream its last work went to; callers
        that overlap in time or use different streams get handles of their own, and their Griffin-Lim calls
        can overlap on the GPU."""
        stream = _lib.current_stream()
        key = stream.value or 0
        with self._handles_mu:
            idle = self._idle.get(key)
            h = idle.pop() if idle else None
        if h is None:
            lib = _lib.load()
            h = ctypes.c_void_p()
            with torch.cuda.device(self._dev()):
                _lib.check(lib.ttsx_vocoder_create(self._stft_params(), stream, ctypes.byref(h)),
                           "ttsx_vocoder_create")
            with self._handles_mu:
                self._handles.append(h)
        try:
            yield h
        finally:
            evict = None
            with self._handles_mu:
                self._idle.setdefault(key, []).append(h)
                # streams come and go with their threads: beyond MAX_IDLE_VOCODERS idle handles, the one that
                # has been idle on another stream the longest is destroyed (destroying waits for its work)
                if sum(len(v) for v in self._idle.values()) > self.MAX_IDLE_VOCODERS:
                    other = next((k for k, v in self._idle.items() if v and k != key), key)
                    evict = self._idle[other].pop(0)
                    if not self._idle[other]:
                        del self._idle[other]
                    self._handles[:] = [x for x in self._handles if x is not evict]
            if evict is not None:
                _lib.load().ttsx_vocoder_destroy(evict)

    def _audio_params(self):
        return _lib.AudioParams(int(bool(self.signal_norm)), int(bool(self.symmetric_norm)),
                                int(bool(self.clip_norm)), float(self.max_norm), float(self.min_level_db),
                                float(self.ref_level_db), float(self.power))

    def inv_mel_basis_device(self):
        if self._inv_basis_dev is None:
            inv = np.linalg.pinv(self._build_mel_basis()).astype(np.float32)
            self._inv_basis_dev = torch.from_numpy(np.ascontiguousarray(inv)).to(self._dev())
        return self._inv_basis_dev

    # ---- batched device API (extension) -------------------------------------
    def spec_to_mag(self, spec, mel=True):
        """spec [B,F,C] normalised (device tensor) -> Griffin-Lim target magnitudes [B,F,num_freq]."""
        lib = _lib.load()
        spec = spec.contiguous().float()
        B, Fr, C = spec.shape
        mag = torch.empty(B, Fr, self.num_freq, device=spec.device, dtype=torch.float32)
        inv = self.inv_mel_basis_device() if mel else None
        ap = self._audio_params()
        with torch.cuda.device(spec.device):
            _lib.check(lib.ttsx_spec_to_mag(_lib.ptr(spec), _lib.ptr(inv), B, Fr, C, self.num_freq, ap,
                                            _lib.ptr(mag), _lib.current_stream()), "ttsx_spec_to_mag")
        return mag

    def griffin_lim_batch(self, mag, init_phase, n_frames=None, deemphasis=True):
        """mag, init_phase [B,F,num_freq] device tensors -> wav [B, hop*(F-1)] float32 (device)."""
        lib = _lib.load()
        mag, init_phase = mag.contiguous().float(), init_phase.contiguous().float()
        B, Fr, _ = mag.shape
        wav = torch.empty(B, self.hop_length * (Fr - 1), device=mag.device, dtype=torch.float32)
        if n_frames is not None:
            n_frames = n_frames.to(device=mag.device, dtype=torch.int32).contiguous()
        pre = float(self.preemphasis) if (deemphasis and self.preemphasis != 0) else 0.0
        if mag.device != self._dev():
            raise ValueError("magnitudes live on %s, this AudioProcessor on %s" % (mag.device, self._dev()))
        with torch.cuda.device(mag.device), self._vocoder() as voc:
            _lib.check(lib.ttsx_vocoder_griffin_lim(voc, _lib.ptr(mag), _lib.ptr(init_phase),
                                                    _lib.ptr(n_frames), B, Fr, int(self.griffin_lim_iters), pre,
                                                    _lib.ptr(wav), _lib.current_stream()), "ttsx_vocoder_griffin_lim")
        return wav

    def request_tail(self, wav, n_samples=None, gap=10000, trim=True, threshold_db=-40, min_silence_sec=0.8):
        """Device-side tail of a request: wav [B,L] float32 (device) -> (pcm int16 ndarray, endpoints
        [B] ndarray, peak).  Every sentence is cut at `find_endpoint`, followed by `gap` zeros, and
        the concatenation is scaled as `save_wav` does (Tacotron2/TTS/synthesizer.py:166-173)."""
        lib = _lib.load()
        wav = wav.contiguous().float()
        B, L = wav.shape
        dev = wav.device
        if n_samples is not None:
            n_samples = torch.as_tensor(n_samples).to(device=dev, dtype=torch.int32).contiguous()
        window = int(self.sample_rate * min_silence_sec)
        hop = int(window / 4)
        cap = B * (L + gap)
        endpoints = torch.empty(B, device=dev, dtype=torch.int32)
        offsets = torch.empty(B + 2, device=dev, dtype=torch.int32)
        pcm = torch.empty(cap, device=dev, dtype=torch.int16)
        total, peak = ctypes.c_long(0), ctypes.c_float(0.0)
        with torch.cuda.device(dev):
            _lib.check(lib.ttsx_request_tail(_lib.ptr(wav), _lib.ptr(n_samples), B, L, window, hop,
                                             float(self._db_to_amp(threshold_db)), int(gap), 1 if trim else 0,
                                             _lib.ptr(endpoints), _lib.ptr(offsets), _lib.ptr(pcm), cap,
                                             ctypes.byref(total), ctypes.byref(peak), _lib.current_stream()),
                       "ttsx_request_tail")
        return pcm[:total.value].cpu().numpy(), endpoints.cpu().numpy(), float(peak.value)

    # ---- reference-shaped API (NumPy in, NumPy out) --------------------------
    def _phase(self, shape, init_phase):
        if init_phase is None:
            init_phase = np.random.rand(*shape)
        return np.asarray(init_phase, dtype=np.float32)

    def _gl_from_mag(self, mag_dev, shape_fk, init_phase, deemphasis):
        ph = self._phase(shape_fk, init_phase)                       # [num_freq, F]
        ph_dev = torch.from_numpy(np.ascontiguousarray(ph.T)).to(self._dev()).unsqueeze(0)
        wav = self.griffin_lim_batch(mag_dev, ph_dev, None, deemphasis)
        return wav[0].cpu().numpy()

    def _griffin_lim(self, S, init_phase=None):
        """S [num_freq, F] magnitudes -> waveform [hop*(F-1)] float32 (utils.py:922-929)."""
        S = np.abs(np.asarray(S)).astype(np.float32)
        mag = torch.from_numpy(np.ascontiguousarray(S.T)).to(self._dev()).unsqueeze(0)
        return self._gl_from_mag(mag, S.shape, init_phase, deemphasis=False)

    def inv_mel_spectrogram(self, mel_spectrogram, init_phase=None):
        """mel [num_mels, F] normalised -> waveform (utils.py:905-912); float64 like the
        reference when de-emphasis is on (lfilter output)."""
        mel = np.asarray(mel_spectrogram, dtype=np.float32)
        spec = torch.from_numpy(np.ascontiguousarray(mel.T)).to(self._dev()).unsqueeze(0)
        mag = self.spec_to_mag(spec, mel=True)
        wav = self._gl_from_mag(mag, (self.num_freq, mel.shape[1]), init_phase, deemphasis=True)
        return wav.astype(np.float64) if self.preemphasis != 0 else wav

    def inv_spectrogram(self, spectrogram, init_phase=None):
        """linear spectrogram [num_freq, F] normalised -> waveform (utils.py:896-903)."""
        spec_np = np.asarray(spectrogram, dtype=np.float32)
        spec = torch.from_numpy(np.ascontiguousarray(spec_np.T)).to(self._dev()).unsqueeze(0)
        mag = self.spec_to_mag(spec, mel=False)
        wav = self._gl_from_mag(mag, spec_np.shape, init_phase, deemphasis=True)
        return wav.astype(np.float64) if self.preemphasis != 0 else wav

    def _stft(self, y):
        """y [hop*(F-1)] -> complex64 [num_freq, F] (librosa.stft semantics, utils.py:931-938)."""
        lib = _lib.load()
        y = np.asarray(y, dtype=np.float32)
        if len(y) % self.hop_length != 0:
            raise ValueError("the GPU STFT handles signals of hop*(F-1) samples (the Griffin-Lim case)")
        Fr = len(y) // self.hop_length + 1
        yd = torch.from_numpy(y).to(self._dev()).unsqueeze(0).contiguous()
        re = torch.empty(1, Fr, self.num_freq, device=yd.device)
        im = torch.empty_like(re)
        with torch.cuda.device(yd.device), self._vocoder() as voc:
            _lib.check(lib.ttsx_vocoder_stft(voc, _lib.ptr(yd), 1, Fr, _lib.ptr(re), _lib.ptr(im),
                                             _lib.current_stream()), "ttsx_vocoder_stft")
        return (re[0].cpu().numpy() + 1j * im[0].cpu().numpy()).T.astype(np.complex64)

    def _istft(self, Y):
        """complex [num_freq, F] -> float32 [hop*(F-1)] (librosa.istft semantics, utils.py:940-942)."""
        lib = _lib.load()
        Y = np.asarray(Y)
        Fr = Y.shape[1]
        re = torch.from_numpy(np.ascontiguousarray(Y.real.T.astype(np.float32))).to(self._dev()).unsqueeze(0)
        im = torch.from_numpy(np.ascontiguousarray(Y.imag.T.astype(np.float32))).to(self._dev()).unsqueeze(0)
        y = torch.empty(1, self.hop_length * (Fr - 1), device=re.device)
        with torch.cuda.device(re.device), self._vocoder() as voc:
            _lib.check(lib.ttsx_vocoder_istft(voc, _lib.ptr(re), _lib.ptr(im), 1, Fr, _lib.ptr(y),
                                              _lib.current_stream()), "ttsx_vocoder_istft")
        return y[0].cpu().numpy()


def _destroy_vocoders(handles, mu):
    with mu:
        hs, handles[:] = list(handles), []
    if hs:
        lib = _lib.load()
        for h in hs:
            lib.ttsx_vocoder_destroy(h)


def wav_bytes(ap, wav):
    """What `Synthesizer.tts` hands to Flask: a WAV file in a BytesIO
    (Tacotron2/TTS/synthesizer.py:172-174)."""
    out = io.BytesIO()
    ap.save_wav(np.asarray(wav), out)
    return out
