"""Synthesis helpers with the reference's signatures (Tacotron2/TTS/utils.py:1422-1528).

`text` may be a string or an integer id sequence (list, ndarray or LongTensor; what the tests
use).  Strings go through `tts_amd.text` as in the reference (utils.py:1423-1437): character ids
when `use_phonemes` is false, phoneme ids otherwise - that branch needs a grapheme-to-phoneme
backend (phonemizer + espeak, or `text_to_ids=`).
"""
import numpy as np
import torch

from . import text as text_frontend


def text_to_seqvec(text, CONFIG, use_cuda, text_to_ids=None):
    if isinstance(text, str):
        cleaners = [CONFIG.get("text_cleaner") or "basic_cleaners"]
        if text_to_ids is not None:
            seq = np.asarray(text_to_ids(text), dtype=np.int32)
        elif CONFIG.get("use_phonemes"):
            seq = np.asarray(text_frontend.phoneme_to_sequence(text, cleaners, CONFIG.get("phoneme_language"),
                                                               bool(CONFIG.get("enable_eos_bos_chars"))), dtype=np.int32)
        else:
            seq = np.asarray(text_frontend.text_to_sequence(text, cleaners), dtype=np.int32)
    else:
        seq = np.asarray(text, dtype=np.int32).reshape(-1)
    chars_var = torch.from_numpy(seq).unsqueeze(0)
    if use_cuda:
        chars_var = chars_var.cuda()
    return chars_var.long()


def run_model(model, inputs, CONFIG, truncated, speaker_id=None, style_mel=None):
    if CONFIG.get("use_gst"):
        raise NotImplementedError("GST is outside the accelerated path")
    if truncated:
        return model.inference_truncated(inputs, speaker_ids=speaker_id)
    return model.inference(inputs, speaker_ids=speaker_id)


def parse_outputs(postnet_output, decoder_output, alignments):
    postnet_output = postnet_output[0].data.cpu().numpy()
    decoder_output = decoder_output[0].data.cpu().numpy()
    alignment = alignments[0].cpu().data.numpy()
    return postnet_output, decoder_output, alignment


def trim_silence(wav, ap):
    return wav[:ap.find_endpoint(wav)]


def inv_spectrogram(postnet_output, ap, CONFIG, init_phase=None):
    if CONFIG.model in ["Tacotron", "TacotronGST"]:
        return ap.inv_spectrogram(postnet_output.T, init_phase=init_phase)
    return ap.inv_mel_spectrogram(postnet_output.T, init_phase=init_phase)


def id_to_torch(speaker_id):
    if speaker_id is not None:
        speaker_id = torch.from_numpy(np.asarray(speaker_id)).unsqueeze(0)
    return speaker_id


def synthesis(model, text, CONFIG, use_cuda, ap, speaker_id=None, style_wav=None, truncated=False,
              enable_eos_bos_chars=False, do_trim_silence=False, text_to_ids=None, init_phase=None):
    """Same contract as the reference `synthesis` (utils.py:1482-1528): returns
    (wav, alignment, decoder_output, postnet_output, stop_tokens)."""
    if CONFIG.model == "TacotronGST" and style_wav is not None:
        raise NotImplementedError("GST is outside the accelerated path")
    inputs = text_to_seqvec(text, CONFIG, use_cuda, text_to_ids)
    speaker_id = id_to_torch(speaker_id)
    decoder_output, postnet_output, alignments, stop_tokens = run_model(
        model, inputs, CONFIG, truncated, speaker_id, None)
    postnet_output, decoder_output, alignment = parse_outputs(postnet_output, decoder_output, alignments)
    wav = inv_spectrogram(postnet_output, ap, CONFIG, init_phase)
    if do_trim_silence:
        wav = trim_silence(wav, ap)
    return wav, alignment, decoder_output, postnet_output, stop_tokens


class Synthesizer(object):
    """Per-sentence loop of the reference server (Tacotron2/TTS/synthesizer.py:141-174) on
    id sequences: inference -> inv_mel_spectrogram -> endpoint trim -> concat with 10 000 zeros."""

    def __init__(self, model, ap, config):
        self.model, self.ap, self.tts_config = model, ap, config

    def tts_ids(self, sentences_ids, init_phases=None):
        from .audio import wav_bytes
        wavs = []
        for i, ids in enumerate(sentences_ids):
            inputs = text_to_seqvec(ids, self.tts_config, True)
            decoder_output, postnet_output, alignments, _ = run_model(self.model, inputs, self.tts_config, False)
            postnet_output, _, _ = parse_outputs(postnet_output, decoder_output, alignments)
            ph = None if init_phases is None else np.asarray(init_phases[i])[:, :postnet_output.shape[0]]
            wav = inv_spectrogram(postnet_output, self.ap, self.tts_config, ph)
            wav = trim_silence(wav, self.ap)
            wavs += list(wav)
            wavs += [0] * 10000
        return wav_bytes(self.ap, np.asarray(wavs))

    def tts_ids_batched(self, sentences_ids, init_phases=None):
        """The same request as `tts_ids`, with all its sentences decoded as ONE padded batch
        (per-utterance stop and length masks make every row equal to its own batch-1 run), one
        batched Griffin-Lim, and the endpoint trim / 10 000-sample gaps / int16 scaling done on the
        device (`ttsx_request_tail`).  Returns the WAV file as a BytesIO like `tts_ids`.  Tacotron2 only."""
        import io
        import scipy.io.wavfile
        lens = [len(ids) for ids in sentences_ids]
        ids = np.zeros((len(lens), max(lens)), dtype=np.int64)
        for b, seq in enumerate(sentences_ids):
            ids[b, :lens[b]] = np.asarray(seq, dtype=np.int64)
        out = self.model.inference_batch(torch.as_tensor(ids), text_lengths=torch.as_tensor(lens))
        post = out["postnet_output"]
        n_frames = (out["n_steps"] * self.model.decoder.r).to(torch.int32)
        B, Fm, _ = post.shape
        nf = n_frames.cpu().numpy()
        phase = torch.zeros(B, Fm, self.ap.num_freq, device=post.device)
        for b in range(B):
            ph = np.random.rand(self.ap.num_freq, int(nf[b])) if init_phases is None else np.asarray(init_phases[b])[:, :int(nf[b])]
            phase[b, :nf[b]] = torch.from_numpy(np.ascontiguousarray(ph.T.astype(np.float32)))
        wav = self.ap.griffin_lim_batch(self.ap.spec_to_mag(post), phase, n_frames)
        n_samples = (n_frames.clamp(min=1) - 1) * self.ap.hop_length
        pcm, self.last_endpoints, _ = self.ap.request_tail(wav, n_samples)
        buf = io.BytesIO()
        scipy.io.wavfile.write(buf, self.ap.sample_rate, pcm)
        return buf
