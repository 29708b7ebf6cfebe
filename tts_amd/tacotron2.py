"""Tacotron2 with the reference's Python surface, running on libttsx (gfx950).

Mirrors `Tacotron2` of the reference (Tacotron2/TTS/models.py:727-813) for the
serving path: constructor arguments, `load_state_dict`, `cuda`, `eval`,
`decoder.max_decoder_steps`, `decoder.set_r`, `decoder.r` and
`inference(text, speaker_ids=None)` with the same return tuple.  The decoder
loop, attention, postnet and the encoder (3 conv blocks, BiLSTM) run in
hand-written HIP kernels through the C ABI; only the embedding lookup is a
stock PyTorch-ROCm op.

The reference's optional variants are available too (all off in the repo's
config.json): attention windowing, forward attention with transition agent and
masking, attention without location features, Graves attention
(`attn_type="graves"`), speaker embeddings (`num_speakers > 1`) and
`inference_truncated`.  The bidirectional decoder is training-only and absent.

Extensions the reference does not have (its `inference` is batch-1 only,
models.py:661): `inference_batch` decodes B utterances at once with
per-utterance stop handling and optional length masking.
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib


def _dev_tensor(v, device):
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(np.ascontiguousarray(v))
    return v.detach().to(device=device, dtype=torch.float32).contiguous()


class _Decoder:
    """The attributes of the reference `Decoder` that callers touch
    (Tacotron2/TTS/models.py:474-483, 527-528; Synthesizer.load_tts sets
    max_decoder_steps and calls set_r, Tacotron2/TTS/synthesizer.py:68-70)."""

    def __init__(self, r):
        self.r_init = r
        self.r = r
        self.memory_dim = 80
        self.max_decoder_steps = 1000
        self.stop_threshold = 0.7

    def set_r(self, new_r):
        self.r = new_r


class Tacotron2:
    def __init__(self, num_chars, num_speakers, r, postnet_output_dim=80, decoder_output_dim=80,
                 attn_type="original", attn_win=False, attn_norm="softmax", prenet_type="original",
                 prenet_dropout=True, forward_attn=False, trans_agent=False, forward_attn_mask=False,
                 location_attn=True, attn_K=5, separate_stopnet=True, bidirectional_decoder=False):
        if attn_type not in ("original", "graves"):
            raise RuntimeError(" [!] Given Attention Type '{attn_type}' is not exist.")
        if attn_norm not in ("softmax", "sigmoid"):
            raise ValueError("Unknown value for attention norm type")
        if prenet_type not in ("original", "bn"):
            raise ValueError("unknown prenet_type %r" % (prenet_type,))
        if bidirectional_decoder:
            # the backward decoder only runs in Tacotron2.forward (training), models.py:797-799
            raise NotImplementedError("bidirectional_decoder is training-only; the accelerated path is inference")
        if postnet_output_dim != 80 or decoder_output_dim != 80:
            raise NotImplementedError("only 80 mel bins are supported")
        if attn_type == "graves" and not 1 <= attn_K <= 16:
            raise ValueError("attn_K must be in 1..16")
        self.num_chars = num_chars
        self.num_speakers = num_speakers
        self.n_frames_per_step = r
        self.attn_norm = attn_norm
        self.prenet_type = prenet_type
        self.attn_type = attn_type
        self.attn_K = attn_K
        self.attn_win = bool(attn_win)
        self.forward_attn = bool(forward_attn)
        self.trans_agent = bool(trans_agent)
        self.forward_attn_mask = bool(forward_attn_mask)
        self.location_attn = bool(location_attn)
        self._rnn_state = None          # encoder BiLSTM state carried by inference_truncated
        self._memory_truncated = None   # frame fed to the first step of the next truncated call
        self.decoder = _Decoder(r)
        self.device = torch.device("cpu")
        self.training = False
        self._sd = None
        self._dec_handle = None
        self._post_handle = None
        self._enc_handle = None
        self._lstm_handle = None

    # ---- torch.nn.Module look-alikes ------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        sd = {}
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            sd[k] = v
        need = ["embedding.weight", "decoder.attention_rnn.weight_ih", "decoder.decoder_rnn.weight_ih",
                "decoder.linear_projection.linear_layer.weight", "postnet.convolutions.4.net.0.weight"]
        if self.attn_type == "graves":
            need += ["decoder.attention.N_a.0.weight", "decoder.attention.N_a.2.weight"]
        else:
            need += ["decoder.attention.query_layer.linear_layer.weight", "decoder.attention.v.linear_layer.weight"]
            if self.location_attn:
                need.append("decoder.attention.location_layer.location_conv.weight")
            if self.forward_attn and self.trans_agent:
                need.append("decoder.attention.ta.weight")
        if self.num_speakers > 1:
            need.append("speaker_embedding.weight")
        missing = [k for k in need if k not in sd]
        if missing:
            raise RuntimeError("Missing key(s) in state_dict: %s" % ", ".join(missing))
        rows = sd["decoder.linear_projection.linear_layer.weight"].shape[0]
        if rows != 80 * self.decoder.r_init:
            raise RuntimeError("size mismatch for decoder.linear_projection: %d rows, expected %d"
                               % (rows, 80 * self.decoder.r_init))
        self._sd_src = sd
        self._release()
        if self.device.type == "cuda":
            self._upload()
        return self

    def cuda(self, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        if getattr(self, "_sd_src", None) is not None:
            self._upload()
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda":
            return self.cuda(device.index)
        self.device = device
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the accelerated path is inference-only")
        return self

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- weights -----------------------------------------------------------
    def _release(self):
        if self._dec_handle is not None:
            _lib.load().ttsx_t2_decoder_destroy(self._dec_handle)
            self._dec_handle = None
        if self._post_handle is not None:
            _lib.load().ttsx_postnet_destroy(self._post_handle)
            self._post_handle = None
        if self._enc_handle is not None:
            _lib.load().ttsx_encoder_convs_destroy(self._enc_handle)
            self._enc_handle = None
        if self._lstm_handle is not None:
            _lib.load().ttsx_encoder_lstm_destroy(self._lstm_handle)
            self._lstm_handle = None

    def _upload(self):
        lib = _lib.load()
        self._release()
        dev = self.device
        sd = {k: _dev_tensor(v, dev) for k, v in self._sd_src.items()}
        self._sd = sd
        bn = self.prenet_type == "bn"
        w = _lib.T2DecoderWeights()
        for i in range(2):
            p = "decoder.prenet.layers.%d." % i
            w.prenet_w[i] = sd[p + "linear_layer.weight"].data_ptr()
            w.prenet_bn_weight[i] = sd[p + "bn.weight"].data_ptr() if bn else None
            w.prenet_bn_bias[i] = sd[p + "bn.bias"].data_ptr() if bn else None
            w.prenet_bn_mean[i] = sd[p + "bn.running_mean"].data_ptr() if bn else None
            w.prenet_bn_var[i] = sd[p + "bn.running_var"].data_ptr() if bn else None
        g = lambda k: sd["decoder." + k].data_ptr()
        graves = self.attn_type == "graves"
        w.attn_rnn_w_ih, w.attn_rnn_w_hh = g("attention_rnn.weight_ih"), g("attention_rnn.weight_hh")
        w.attn_rnn_b_ih, w.attn_rnn_b_hh = g("attention_rnn.bias_ih"), g("attention_rnn.bias_hh")
        if not graves:
            w.query_w = g("attention.query_layer.linear_layer.weight")
            w.inputs_w = g("attention.inputs_layer.linear_layer.weight")
            w.v_w, w.v_b = g("attention.v.linear_layer.weight"), g("attention.v.linear_layer.bias")
            if self.location_attn:
                w.loc_conv_w = g("attention.location_layer.location_conv.weight")
                w.loc_dense_w = g("attention.location_layer.location_dense.linear_layer.weight")
        w.dec_rnn_w_ih, w.dec_rnn_w_hh = g("decoder_rnn.weight_ih"), g("decoder_rnn.weight_hh")
        w.dec_rnn_b_ih, w.dec_rnn_b_hh = g("decoder_rnn.bias_ih"), g("decoder_rnn.bias_hh")
        w.proj_w, w.proj_b = g("linear_projection.linear_layer.weight"), g("linear_projection.linear_layer.bias")
        w.stop_w, w.stop_b = g("stopnet.1.linear_layer.weight"), g("stopnet.1.linear_layer.bias")
        handle = ctypes.c_void_p()
        norm = _lib.NORM_SIGMOID if self.attn_norm == "sigmoid" else _lib.NORM_SOFTMAX
        with torch.cuda.device(dev):
            _lib.check(lib.ttsx_t2_decoder_create(ctypes.byref(w), self.decoder.r_init, norm,
                                                  _lib.current_stream(), ctypes.byref(handle)),
                       "ttsx_t2_decoder_create")
            self._dec_handle = handle
            if graves or self.attn_win or self.forward_attn:
                o = _lib.T2AttentionOptions()
                o.windowing, o.forward_attn = int(self.attn_win), int(self.forward_attn)
                o.forward_attn_mask = int(self.forward_attn and self.forward_attn_mask)
                o.trans_agent = int(self.forward_attn and self.trans_agent)
                if o.trans_agent:
                    o.ta_w, o.ta_b = g("attention.ta.weight"), g("attention.ta.bias")
                if graves:
                    o.graves_K = self.attn_K
                    o.graves_w1, o.graves_b1 = g("attention.N_a.0.weight"), g("attention.N_a.0.bias")
                    o.graves_w2, o.graves_b2 = g("attention.N_a.2.weight"), g("attention.N_a.2.bias")
                _lib.check(lib.ttsx_t2_decoder_set_attention(handle, ctypes.byref(o), _lib.current_stream()),
                           "ttsx_t2_decoder_set_attention")
            pw = _lib.PostnetWeights()
            for i in range(5):
                p = "postnet.convolutions.%d.net." % i
                pw.conv_w[i], pw.conv_b[i] = sd[p + "0.weight"].data_ptr(), sd[p + "0.bias"].data_ptr()
                pw.bn_weight[i], pw.bn_bias[i] = sd[p + "1.weight"].data_ptr(), sd[p + "1.bias"].data_ptr()
                pw.bn_mean[i], pw.bn_var[i] = sd[p + "1.running_mean"].data_ptr(), sd[p + "1.running_var"].data_ptr()
            ph = ctypes.c_void_p()
            _lib.check(lib.ttsx_postnet_create(ctypes.byref(pw), _lib.current_stream(), ctypes.byref(ph)),
                       "ttsx_postnet_create")
            self._post_handle = ph
            ew = _lib.EncoderConvWeights()
            for i in range(3):
                p = "encoder.convolutions.%d.net." % i
                ew.conv_w[i], ew.conv_b[i] = sd[p + "0.weight"].data_ptr(), sd[p + "0.bias"].data_ptr()
                ew.bn_weight[i], ew.bn_bias[i] = sd[p + "1.weight"].data_ptr(), sd[p + "1.bias"].data_ptr()
                ew.bn_mean[i], ew.bn_var[i] = sd[p + "1.running_mean"].data_ptr(), sd[p + "1.running_var"].data_ptr()
            eh = ctypes.c_void_p()
            _lib.check(lib.ttsx_encoder_convs_create(ctypes.byref(ew), _lib.current_stream(), ctypes.byref(eh)),
                       "ttsx_encoder_convs_create")
            self._enc_handle = eh
            lw = _lib.EncoderLstmWeights()
            for d, sfx in enumerate(("_l0", "_l0_reverse")):
                lw.w_ih[d], lw.w_hh[d] = sd["encoder.lstm.weight_ih" + sfx].data_ptr(), sd["encoder.lstm.weight_hh" + sfx].data_ptr()
                lw.b_ih[d], lw.b_hh[d] = sd["encoder.lstm.bias_ih" + sfx].data_ptr(), sd["encoder.lstm.bias_hh" + sfx].data_ptr()
            lh = ctypes.c_void_p()
            _lib.check(lib.ttsx_encoder_lstm_create(ctypes.byref(lw), _lib.current_stream(), ctypes.byref(lh)),
                       "ttsx_encoder_lstm_create")
            self._lstm_handle = lh

    def _require_device(self):
        if self.device.type != "cuda" or self._dec_handle is None:
            raise _lib.TtsxError("Tacotron2 runs on the GPU only: call .cuda() after load_state_dict() "
                                 "(there is no CPU path)")

    # ---- forward pieces ----------------------------------------------------
    @torch.no_grad()
    def encode(self, text, text_lengths=None, speaker_ids=None, rnn_state=None):
        """Embedding + Encoder.inference (Tacotron2/TTS/models.py:803-804, 449-454).  The three
        conv blocks (position-major, so the embedding output is consumed as it is) and the BiLSTM
        run in HIP kernels; the embedding lookup is a stock op.  With `text_lengths` every
        utterance is encoded as its unpadded batch-1 run would be.  `speaker_ids` adds the
        speaker embedding (_add_speaker_embedding, :838-849); `rnn_state` = (h, c), each
        [2,B,256], is the BiLSTM state to start from and is overwritten with the final one
        (Encoder.inference_truncated, :456-464)."""
        lib = _lib.load()
        sd = self._sd
        x = F.embedding(text, sd["embedding.weight"]).contiguous()          # [B,T,512]
        B, T, _ = x.shape
        y = torch.empty_like(x)
        lens = None
        if text_lengths is not None:
            lens = torch.as_tensor(text_lengths).to(device=x.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.ttsx_encoder_convs_run(self._enc_handle, _lib.ptr(x), _lib.ptr(lens), B, T, _lib.ptr(y),
                                                  _lib.current_stream()), "ttsx_encoder_convs_run")
            if rnn_state is None:
                _lib.check(lib.ttsx_encoder_lstm_run(self._lstm_handle, _lib.ptr(y), _lib.ptr(lens), B, T,
                                                     _lib.ptr(out), _lib.current_stream()), "ttsx_encoder_lstm_run")
            else:
                _lib.check(lib.ttsx_encoder_lstm_run_stateful(self._lstm_handle, _lib.ptr(y), _lib.ptr(lens), B, T,
                                                              _lib.ptr(out), _lib.ptr(rnn_state[0]),
                                                              _lib.ptr(rnn_state[1]), _lib.current_stream()),
                           "ttsx_encoder_lstm_run_stateful")
            if self.num_speakers > 1:
                if speaker_ids is None:
                    raise RuntimeError(" [!] Model has speaker embedding layer but speaker_id is not provided")
                ids = torch.as_tensor(speaker_ids).reshape(-1).cpu()
                if ids.numel() != B or int(ids.min()) < 0 or int(ids.max()) >= self.num_speakers:
                    raise IndexError("speaker_ids must hold one id in [0, %d) per utterance" % self.num_speakers)
                ids = ids.to(device=x.device, dtype=torch.int32).contiguous()
                _lib.check(lib.ttsx_add_speaker_embedding(_lib.ptr(out), _lib.ptr(sd["speaker_embedding.weight"]),
                                                          _lib.ptr(ids), self.num_speakers, B, T,
                                                          _lib.current_stream()), "ttsx_add_speaker_embedding")
        return out

    @torch.no_grad()
    def decode(self, enc, enc_len=None, fixed_steps=None, keep_states=False, memory=None):
        """Greedy decoder loop on the GPU.  Returns (mel [B,F,80], align [B,S,T], stop [B,S],
        n_steps [B] int32, steps_run).  `keep_states` / `memory` [B,80]: start from the recurrent
        state the previous call left and feed `memory` to the first step (inference_truncated)."""
        lib = _lib.load()
        B, T, _ = enc.shape
        r = self.decoder.r
        max_steps = int(fixed_steps) if fixed_steps else int(self.decoder.max_decoder_steps)
        dev = enc.device
        mel = torch.empty(B, max_steps * r, 80, device=dev, dtype=torch.float32)
        align = torch.empty(B, max_steps, T, device=dev, dtype=torch.float32)
        stop = torch.empty(B, max_steps, device=dev, dtype=torch.float32)
        n_steps = torch.empty(B, device=dev, dtype=torch.int32)
        steps_run = ctypes.c_int(0)
        if enc_len is not None:
            enc_len = enc_len.to(device=dev, dtype=torch.int32).contiguous()
        with torch.cuda.device(dev):
            if keep_states or memory is not None:
                if memory is not None:
                    memory = memory.to(device=dev, dtype=torch.float32).contiguous()
                _lib.check(lib.ttsx_t2_decoder_set_initial(self._dec_handle, 1 if keep_states else 0,
                                                           _lib.ptr(memory), B, _lib.current_stream()),
                           "ttsx_t2_decoder_set_initial")
            _lib.check(lib.ttsx_t2_decoder_run(self._dec_handle, _lib.ptr(enc), _lib.ptr(enc_len), B, T, r,
                                               max_steps, self.decoder.stop_threshold,
                                               1 if fixed_steps else 0, _lib.ptr(mel), _lib.ptr(align),
                                               _lib.ptr(stop), _lib.ptr(n_steps), ctypes.byref(steps_run),
                                               _lib.current_stream()), "ttsx_t2_decoder_run")
        s = steps_run.value
        return mel[:, :s * r], align[:, :s], stop[:, :s], n_steps, s

    def set_plan(self, fused_attention=-1, split_lstm=-1, share_query=-1):
        """Override the decoder's launch plan (ttsx_t2_decoder_set_plan): -1 by shape, 0 off, 1 on."""
        _lib.check(_lib.load().ttsx_t2_decoder_set_plan(self._dec_handle, int(fused_attention), int(split_lstm),
                                                        int(share_query)), "ttsx_t2_decoder_set_plan")

    def set_fusion(self, prenet_in_projection=-1):
        """Batched decode: 1 = prenet layer 2 inside the previous step's projection launch, 2 = also the late decoder-LSTM
        columns inside the attention launch, 0 = neither, -1 = by shape (ttsx_t2_decoder_set_fusion)."""
        _lib.check(_lib.load().ttsx_t2_decoder_set_fusion(self._dec_handle, int(prenet_in_projection)),
                   "ttsx_t2_decoder_set_fusion")

    def set_encoder_plan(self, scan=-1):
        """Encoder BiLSTM recurrence as one persistent launch (1), as a chain of step launches (0), or by shape (-1)
        (ttsx_encoder_lstm_set_plan)."""
        _lib.check(_lib.load().ttsx_encoder_lstm_set_plan(self._lstm_handle, int(scan)), "ttsx_encoder_lstm_set_plan")

    def set_pieces(self, staged=-1):
        """Padded batch 64: early-column pieces as 32 x 32 register tiles (0) or staged through LDS (1)
        (ttsx_t2_decoder_set_pieces); -1 by shape."""
        _lib.check(_lib.load().ttsx_t2_decoder_set_pieces(self._dec_handle, int(staged)), "ttsx_t2_decoder_set_pieces")

    def set_team(self, team=-1):
        """Whether batch-1 runs may use the persistent team kernel (ttsx_t2_decoder_set_team): -1 by shape, 0 never,
        1 wherever the shapes allow."""
        _lib.check(_lib.load().ttsx_t2_decoder_set_team(self._dec_handle, int(team)), "ttsx_t2_decoder_set_team")

    def last_steps_launched(self):
        """Decoder steps the last decode() enqueued (at least the steps any utterance needed)."""
        return int(_lib.load().ttsx_t2_decoder_steps_launched(self._dec_handle))

    @torch.no_grad()
    def postnet(self, mel, n_frames=None):
        """mel + Postnet(mel) (Tacotron2/TTS/models.py:809-810) on [B,F,80]."""
        lib = _lib.load()
        mel = mel.contiguous()
        B, Fr, _ = mel.shape
        out = torch.empty_like(mel)
        if n_frames is not None:
            n_frames = n_frames.to(device=mel.device, dtype=torch.int32).contiguous()
        with torch.cuda.device(mel.device):
            _lib.check(lib.ttsx_postnet_run(self._post_handle, _lib.ptr(mel), _lib.ptr(n_frames), B, Fr,
                                            _lib.ptr(out), _lib.current_stream()), "ttsx_postnet_run")
        return out

    # ---- public API ----------------------------------------------------------
    @torch.no_grad()
    def inference(self, text, speaker_ids=None):
        """Same contract as the reference `Tacotron2.inference`
        (Tacotron2/TTS/models.py:802-813): text LongTensor [B,T] ->
        (decoder_output [B,F,80], postnet_output [B,F,80], alignments [B,T_dec,T],
        stop_tokens [B,T_dec,1])."""
        out = self.inference_batch(text, speaker_ids=speaker_ids)
        return out["decoder_output"], out["postnet_output"], out["alignments"], out["stop_tokens"]

    @torch.no_grad()
    def inference_truncated(self, text, speaker_ids=None):
        """Same contract as the reference `Tacotron2.inference_truncated`
        (Tacotron2/TTS/models.py:815-829): the encoder BiLSTM state (:456-464), the decoder's
        LSTM states and context (:675-686) and the fed-back frame survive from call to call;
        the attention state is reset per call.  As in the reference the fed-back frame is the
        whole decoder output (:703), which only fits the prenet at r = 1, and it is the frame
        of the step BEFORE the one that ended the previous call (the assignment is skipped on
        `break`)."""
        self._require_device()
        if self.decoder.r != 1:
            raise RuntimeError("inference_truncated feeds the whole decoder output back: it needs r == 1")
        text = torch.as_tensor(text).to(self.device).long()
        if text.dim() != 2 or text.shape[0] != 1:
            raise ValueError("text must be [1, T]")
        first = self._rnn_state is None
        if first:
            self._rnn_state = (torch.zeros(2, 1, 256, device=self.device), torch.zeros(2, 1, 256, device=self.device))
            self._memory_truncated = torch.zeros(1, 80, device=self.device)
        enc = self.encode(text, None, speaker_ids, self._rnn_state)
        mel, align, stop, n_steps, steps = self.decode(enc, None, None, keep_states=not first,
                                                       memory=self._memory_truncated)
        if steps == self.decoder.max_decoder_steps and float(stop[0, -1]) <= self.decoder.stop_threshold:
            print("   | > Decoder stopped with 'max_decoder_steps")
        if steps > 1:
            self._memory_truncated = mel[:, steps - 2].clone()
        post = self.postnet(mel)
        return mel, post, align, stop.unsqueeze(-1)

    def reset_truncated(self):
        """Forget the state `inference_truncated` carries (a fresh reference model)."""
        self._rnn_state = None
        self._memory_truncated = None

    @torch.no_grad()
    def inference_batch(self, text, text_lengths=None, fixed_steps=None, speaker_ids=None):
        """Batched decode.  Each utterance's rows up to n_steps[b] equal what a
        batch-1 `inference` of that utterance returns; later rows are padding."""
        self._require_device()
        text = torch.as_tensor(text).to(self.device).long()
        if text.dim() != 2:
            raise ValueError("text must be [B, T]")
        enc = self.encode(text, text_lengths, speaker_ids)
        mel, align, stop, n_steps, steps = self.decode(enc, text_lengths, fixed_steps)
        if not fixed_steps and steps == self.decoder.max_decoder_steps and \
                bool((stop[:, -1] <= self.decoder.stop_threshold).any()):
            print("   | > Decoder stopped with 'max_decoder_steps")
        n_frames = (n_steps * self.decoder.r) if text.shape[0] > 1 else None
        post = self.postnet(mel, n_frames)
        return dict(decoder_output=mel, postnet_output=post, alignments=align,
                    stop_tokens=stop.unsqueeze(-1), n_steps=n_steps, encoder_outputs=enc)


def setup_model(num_chars, num_speakers, c):
    """Counterpart of `setup_model` (Tacotron2/TTS/utils.py:1267-1286)."""
    print("\t[-] Using model: {}".format(c.model))
    return Tacotron2(num_chars=num_chars, num_speakers=num_speakers, r=c.r,
                     postnet_output_dim=c.audio["num_mels"], decoder_output_dim=c.audio["num_mels"],
                     attn_type=c.attention_type, attn_win=c.windowing, attn_norm=c.attention_norm,
                     prenet_type=c.prenet_type, prenet_dropout=c.prenet_dropout,
                     forward_attn=c.use_forward_attn, trans_agent=c.transition_agent,
                     forward_attn_mask=c.forward_attn_mask, location_attn=c.location_attn,
                     attn_K=c.attention_heads, separate_stopnet=c.separate_stopnet,
                     bidirectional_decoder=c.bidirectional_decoder)
