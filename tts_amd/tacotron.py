"""Tacotron (v1) with the reference's Python surface, running on libttsx (gfx950).

Mirrors `Tacotron` of the reference (Tacotron/TTS/models.py:670-700) for the
serving path: constructor arguments, `load_state_dict`, `cuda`, `eval`,
`decoder.max_decoder_steps` and `forward(characters)` with the same return
tuple (mel_outputs, linear_outputs, alignments, stop_tokens).  Prenets, both
CBHGs, the greedy attention decoder and last_linear run in hand-written HIP
kernels through the C ABI; only the embedding lookup is a stock op.

The reference decoder's stop rule calls `.item()` on a one-element tensor
(models.py:635), so its `forward` is batch-1 only; here every utterance of a
batch stops on its own and rows past `n_steps[b]` are padding.
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
    (Tacotron/TTS/models.py:496-503)."""

    def __init__(self, r, memory_size):
        self.r = r
        self.memory_size = memory_size if memory_size > 0 else r
        self.memory_dim = 80
        self.max_decoder_steps = 500


class Tacotron:
    def __init__(self, num_chars, embedding_dim=256, linear_dim=1025, mel_dim=80, r=5, padding_idx=None,
                 memory_size=5, attn_windowing=False):
        if embedding_dim != 256 or mel_dim != 80:
            raise NotImplementedError("only embedding_dim=256 and mel_dim=80 are supported")
        if attn_windowing:
            raise NotImplementedError("attention windowing is outside the accelerated path")
        self.num_chars = num_chars
        self.r = r
        self.mel_dim = mel_dim
        self.linear_dim = linear_dim
        self.decoder = _Decoder(r, memory_size)
        self.device = torch.device("cpu")
        self.training = False
        self._sd_src = None
        self._sd = None
        self._handle = None

    # ---- torch.nn.Module look-alikes ------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        need = ["embedding.weight", "encoder.cbhg.cbhg.gru.weight_hh_l0", "decoder.proj_to_mel.weight",
                "decoder.memory_init.weight", "postnet.cbhg.pre_highway.weight", "last_linear.0.weight"]
        missing = [k for k in need if k not in sd]
        if missing:
            raise RuntimeError("Missing key(s) in state_dict: %s" % ", ".join(missing))
        rows = sd["decoder.proj_to_mel.weight"].shape[0]
        if rows != 80 * self.r:
            raise RuntimeError("size mismatch for decoder.proj_to_mel: %d rows, expected %d" % (rows, 80 * self.r))
        cols = sd["decoder.memory_init.weight"].shape[1]
        if cols != 80 * self.decoder.memory_size:
            raise RuntimeError("size mismatch for decoder.memory_init: %d columns, expected %d"
                               % (cols, 80 * self.decoder.memory_size))
        self._sd_src = sd
        self._release()
        if self.device.type == "cuda":
            self._upload()
        return self

    def cuda(self, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        if self._sd_src is not None:
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

    def _release(self):
        if self._handle is not None:
            _lib.load().ttsx_t1_destroy(self._handle)
            self._handle = None

    @staticmethod
    def _fill_cbhg(cw, sd, prefix, K, in_features, proj_dims, pre_highway):
        cw.K, cw.in_features = K, in_features
        cw.proj_dims[0], cw.proj_dims[1] = proj_dims
        p = lambda k: sd[prefix + k].data_ptr()
        for k in range(K):
            b = "conv1d_banks.%d." % k
            cw.bank_w[k] = p(b + "conv1d.weight")
            cw.bank_bn_weight[k], cw.bank_bn_bias[k] = p(b + "bn.weight"), p(b + "bn.bias")
            cw.bank_bn_mean[k], cw.bank_bn_var[k] = p(b + "bn.running_mean"), p(b + "bn.running_var")
        for i in range(2):
            b = "conv1d_projections.%d." % i
            cw.proj_w[i] = p(b + "conv1d.weight")
            cw.proj_bn_weight[i], cw.proj_bn_bias[i] = p(b + "bn.weight"), p(b + "bn.bias")
            cw.proj_bn_mean[i], cw.proj_bn_var[i] = p(b + "bn.running_mean"), p(b + "bn.running_var")
        cw.pre_highway_w = p("pre_highway.weight") if pre_highway else None
        for i in range(4):
            cw.hw_H_w[i], cw.hw_H_b[i] = p("highways.%d.H.weight" % i), p("highways.%d.H.bias" % i)
            cw.hw_T_w[i], cw.hw_T_b[i] = p("highways.%d.T.weight" % i), p("highways.%d.T.bias" % i)
        for d, sfx in enumerate(("_l0", "_l0_reverse")):
            cw.gru_w_ih[d], cw.gru_w_hh[d] = p("gru.weight_ih" + sfx), p("gru.weight_hh" + sfx)
            cw.gru_b_ih[d], cw.gru_b_hh[d] = p("gru.bias_ih" + sfx), p("gru.bias_hh" + sfx)

    def _upload(self):
        lib = _lib.load()
        self._release()
        dev = self.device
        sd = {k: _dev_tensor(v, dev) for k, v in self._sd_src.items()}
        self._sd = sd
        p = lambda k: sd[k].data_ptr()
        w = _lib.T1Weights()
        for i in range(2):
            w.enc_prenet_w[i], w.enc_prenet_b[i] = p("encoder.prenet.layers.%d.weight" % i), p("encoder.prenet.layers.%d.bias" % i)
            w.dec_prenet_w[i], w.dec_prenet_b[i] = p("decoder.prenet.layers.%d.weight" % i), p("decoder.prenet.layers.%d.bias" % i)
        self._fill_cbhg(w.enc_cbhg, sd, "encoder.cbhg.cbhg.", 16, 128, (128, 128), False)
        self._fill_cbhg(w.post_cbhg, sd, "postnet.cbhg.", 8, 80, (256, 80), True)
        a = "decoder.attention_rnn."
        w.att_rnn_w_ih, w.att_rnn_w_hh = p(a + "rnn_cell.weight_ih"), p(a + "rnn_cell.weight_hh")
        w.att_rnn_b_ih, w.att_rnn_b_hh = p(a + "rnn_cell.bias_ih"), p(a + "rnn_cell.bias_hh")
        am = a + "alignment_model."
        w.loc_conv_w = p(am + "loc_conv.1.weight")
        w.loc_linear_w, w.loc_linear_b = p(am + "loc_linear.weight"), p(am + "loc_linear.bias")
        w.query_w, w.query_b = p(am + "query_layer.weight"), p(am + "query_layer.bias")
        w.annot_w, w.annot_b = p(am + "annot_layer.weight"), p(am + "annot_layer.bias")
        w.v_w = p(am + "v.weight")
        w.proj_dec_in_w, w.proj_dec_in_b = p("decoder.project_to_decoder_in.weight"), p("decoder.project_to_decoder_in.bias")
        for i in range(2):
            d = "decoder.decoder_rnns.%d." % i
            w.dec_rnn_w_ih[i], w.dec_rnn_w_hh[i] = p(d + "weight_ih"), p(d + "weight_hh")
            w.dec_rnn_b_ih[i], w.dec_rnn_b_hh[i] = p(d + "bias_ih"), p(d + "bias_hh")
        w.proj_mel_w, w.proj_mel_b = p("decoder.proj_to_mel.weight"), p("decoder.proj_to_mel.bias")
        w.att_rnn_init = p("decoder.attention_rnn_init.weight")
        w.memory_init = p("decoder.memory_init.weight")
        w.dec_rnn_inits = p("decoder.decoder_rnn_inits.weight")
        w.stop_w, w.stop_b = p("decoder.stopnet.linear.weight"), p("decoder.stopnet.linear.bias")
        w.last_linear_w, w.last_linear_b = p("last_linear.0.weight"), p("last_linear.0.bias")
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.ttsx_t1_create(ctypes.byref(w), self.r, self.decoder.memory_size, self.linear_dim,
                                          _lib.current_stream(), ctypes.byref(h)), "ttsx_t1_create")
        self._handle = h

    def _require_device(self):
        if self.device.type != "cuda" or self._handle is None:
            raise _lib.TtsxError("Tacotron runs on the GPU only: call .cuda() after load_state_dict() "
                                 "(there is no CPU path)")

    # ---- forward pieces ----------------------------------------------------
    @torch.no_grad()
    def _lens(self, lengths, B, hi, what):
        if lengths is None:
            return None
        lens = torch.as_tensor(lengths).to(dtype=torch.int32).cpu()
        if lens.shape != (B,) or int(lens.min()) < 1 or int(lens.max()) > hi:
            raise ValueError("%s must be [B] with 1 <= value <= %d" % (what, hi))
        return lens.to(self.device).contiguous()

    @torch.no_grad()
    def encode(self, characters, text_lengths=None):
        """Embedding + Encoder.forward (Tacotron/TTS/models.py:694-695, 459-469).  With
        `text_lengths` every utterance is encoded as its unpadded batch-1 run would be."""
        lib = _lib.load()
        x = F.embedding(characters, self._sd["embedding.weight"]).contiguous()
        B, T, _ = x.shape
        lens = self._lens(text_lengths, B, T, "text_lengths")
        enc = torch.empty(B, T, 256, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(lib.ttsx_t1_encoder_run(self._handle, _lib.ptr(x), _lib.ptr(lens), B, T, _lib.ptr(enc),
                                               _lib.current_stream()), "ttsx_t1_encoder_run")
        return enc

    @torch.no_grad()
    def decode(self, enc, text_lengths=None):
        """Greedy Decoder.forward (models.py:586-645).  Returns (mel [B,S*r,80], align [B,S,T],
        stop [B,S], n_steps [B] int32)."""
        lib = _lib.load()
        B, T, _ = enc.shape
        lens = self._lens(text_lengths, B, T, "text_lengths")
        r, mx = self.r, int(self.decoder.max_decoder_steps)
        dev = enc.device
        mel = torch.empty(B, (mx + 1) * r, 80, device=dev, dtype=torch.float32)
        align = torch.empty(B, mx + 1, T, device=dev, dtype=torch.float32)
        stop = torch.empty(B, mx + 1, device=dev, dtype=torch.float32)
        n_steps = torch.empty(B, device=dev, dtype=torch.int32)
        run = ctypes.c_int(0)
        with torch.cuda.device(dev):
            _lib.check(lib.ttsx_t1_decoder_run(self._handle, _lib.ptr(enc.contiguous()), _lib.ptr(lens), B, T, mx,
                                               _lib.ptr(mel),
                                               _lib.ptr(align), _lib.ptr(stop), _lib.ptr(n_steps), ctypes.byref(run),
                                               _lib.current_stream()), "ttsx_t1_decoder_run")
        s = run.value
        if s > mx:
            print("   | > Decoder stopped with 'max_decoder_steps")
        if B > 1:
            # rows an utterance never wrote are padding: make them zeros rather than stale memory
            live = torch.arange(s, device=dev)[None, :] < n_steps[:, None]
            dead = ~live
            mel[:, :s * r].masked_fill_(dead.repeat_interleave(r, 1)[:, :, None], 0.0)
            align[:, :s].masked_fill_(dead[:, :, None], 0.0)
            stop[:, :s].masked_fill_(dead, 0.0)
        return mel[:, :s * r], align[:, :s], stop[:, :s], n_steps

    def set_plan(self, decoder=-1):
        """Decoder form of the following runs (ttsx_t1_set_plan): -1 by shape, 0 the launch chain, 1 the on-chip
        team kernel (batch 1-2, T <= 128), 2 one workgroup per utterance."""
        _lib.check(_lib.load().ttsx_t1_set_plan(self._handle, int(decoder)), "ttsx_t1_set_plan")

    @torch.no_grad()
    def postnet(self, mel, n_frames=None):
        """sigmoid(last_linear(PostCBHG(mel))) (models.py:698-699) on [B,F,80].  With `n_frames`
        every utterance is processed as its unpadded batch-1 run would be; padding rows are zero."""
        lib = _lib.load()
        mel = mel.contiguous()
        B, Fr, _ = mel.shape
        nf = self._lens(n_frames, B, Fr, "n_frames")
        lin = torch.empty(B, Fr, self.linear_dim, device=mel.device, dtype=torch.float32)
        with torch.cuda.device(mel.device):
            _lib.check(lib.ttsx_t1_postnet_run(self._handle, _lib.ptr(mel), _lib.ptr(nf), B, Fr, _lib.ptr(lin),
                                               _lib.current_stream()), "ttsx_t1_postnet_run")
        if nf is not None:
            lin.masked_fill_((torch.arange(Fr, device=mel.device)[None, :] >= nf[:, None])[:, :, None], 0.0)
        return lin

    # ---- public API ----------------------------------------------------------
    @torch.no_grad()
    def forward(self, characters, mel_specs=None, mask=None):
        """Same contract as the reference `Tacotron.forward` in eval mode
        (Tacotron/TTS/models.py:692-700): characters LongTensor [B,T] ->
        (mel_outputs [B,F,80], linear_outputs [B,F,linear_dim], alignments [B,S,T],
        stop_tokens [B,S,1])."""
        if mel_specs is not None or mask is not None:
            raise NotImplementedError("teacher forcing and masking are training-time paths")
        out = self.forward_batch(characters)
        return out["mel_outputs"], out["linear_outputs"], out["alignments"], out["stop_tokens"]

    __call__ = forward

    @torch.no_grad()
    def forward_batch(self, characters, text_lengths=None):
        """Batched synthesis (the reference's stop rule limits it to one utterance per call,
        models.py:635).  Each utterance's rows up to n_steps[b] equal what a batch-1 `forward`
        of its unpadded text returns; later rows are zero padding."""
        self._require_device()
        characters = torch.as_tensor(characters).to(self.device).long()
        if characters.dim() != 2:
            raise ValueError("characters must be [B, T]")
        enc = self.encode(characters, text_lengths)
        mel, align, stop, n_steps = self.decode(enc, text_lengths)
        lin = self.postnet(mel, n_steps * self.r if characters.shape[0] > 1 else None)
        return dict(mel_outputs=mel, linear_outputs=lin, alignments=align, stop_tokens=stop.unsqueeze(-1),
                    n_steps=n_steps, encoder_outputs=enc)


def setup_model(num_chars, c, ap):
    """The model construction of the Tacotron synthesizer (Tacotron/TTS/synthesizer.py:26-35).
    The reference passes `config.memory_size` positionally, where it lands on `padding_idx`
    (which does not change an eval-mode lookup); the decoder's memory_size therefore stays at
    its default of 5.  Reproduced as is so that the reference's checkpoints load."""
    return Tacotron(num_chars, c.embedding_size, ap.num_freq, ap.num_mels, c.r, c.memory_size)
