"""CPU oracle for the Tacotron2 synthesis path (TEST INFRASTRUCTURE ONLY).

A plain restatement, in torch CPU tensor primitives (matmul / conv1d /
elementwise), of what the reference computes in
Tacotron2/TTS/models.py:802-813 (`Tacotron2.inference`).  Nothing here is
imported by the product path (`tts_amd/`); only tests, `bench.py`'s
`cpu_baseline` leg and `__graft_entry__.smoke()` use it, as the checker.

Pinned against the reference itself: `oracle/gen_golden.py` imports
/root/reference/Tacotron2/TTS/models.py in the build container, runs the same
synthetic checkpoints through both and writes `tests/golden/t2_*.npz`; the
CPU test-suite replays those fixtures against this file.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, key, dtype):
    return torch.as_tensor(np.asarray(sd[key])).to(dtype)


def _bn_eval(x, sd, prefix, dtype, eps=1e-5):
    """BatchNorm1d in eval mode over the channel axis 1."""
    w, b = _t(sd, prefix + ".weight", dtype), _t(sd, prefix + ".bias", dtype)
    m, v = _t(sd, prefix + ".running_mean", dtype), _t(sd, prefix + ".running_var", dtype)
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - m.view(shape)) / torch.sqrt(v.view(shape) + eps) * w.view(shape) + b.view(shape)


def conv_bn_block(x, sd, prefix, act, dtype):
    """ConvBNBlock, Tacotron2/TTS/models.py:380-400 (dropout is identity in eval)."""
    w = _t(sd, prefix + ".net.0.weight", dtype)
    b = _t(sd, prefix + ".net.0.bias", dtype)
    x = F.conv1d(x, w, b, padding=(w.shape[-1] - 1) // 2)
    x = _bn_eval(x, sd, prefix + ".net.1", dtype)
    if act == "relu":
        x = torch.relu(x)
    elif act == "tanh":
        x = torch.tanh(x)
    return x


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell math, gate order i, f, g, o."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def encoder(sd, ids, dtype=torch.float32, rnn_state=None, return_state=False):
    """Embedding + Encoder.inference, Tacotron2/TTS/models.py:803-804, 449-454.
    With `rnn_state` = (h0, c0), each [2,B,256] (index 0 forward, 1 reverse), the BiLSTM starts
    from that state and `return_state` also returns the final one: Encoder.inference_truncated,
    models.py:456-464 (nn.LSTM semantics: the reverse direction starts at the last position from
    its own row of the state and ends at position 0)."""
    emb = _t(sd, "embedding.weight", dtype)
    x = emb[torch.as_tensor(ids, dtype=torch.long)].transpose(1, 2)      # [B,512,T]
    for i in range(3):
        x = conv_bn_block(x, sd, "encoder.convolutions.%d" % i, "relu", dtype)
    x = x.transpose(1, 2)                                                # [B,T,512]
    B, T, _ = x.shape
    outs, hn, cn = [], [], []
    for d, (sfx, order) in enumerate((("_l0", range(T)), ("_l0_reverse", range(T - 1, -1, -1)))):
        w_ih = _t(sd, "encoder.lstm.weight_ih" + sfx, dtype)
        w_hh = _t(sd, "encoder.lstm.weight_hh" + sfx, dtype)
        b_ih = _t(sd, "encoder.lstm.bias_ih" + sfx, dtype)
        b_hh = _t(sd, "encoder.lstm.bias_hh" + sfx, dtype)
        h = torch.zeros(B, 256, dtype=dtype)
        c = torch.zeros(B, 256, dtype=dtype)
        if rnn_state is not None:
            h, c = rnn_state[0][d].to(dtype), rnn_state[1][d].to(dtype)
        seq = [None] * T
        for t in order:
            h, c = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            seq[t] = h
        outs.append(torch.stack(seq, dim=1))
        hn.append(h), cn.append(c)
    out = torch.cat(outs, dim=2)                                         # [B,T,512]
    if return_state:
        return out, (torch.stack(hn), torch.stack(cn))
    return out


def add_speaker_embedding(sd, enc, speaker_ids, dtype=torch.float32):
    """Tacotron2._add_speaker_embedding, Tacotron2/TTS/models.py:838-849: the speaker's row of
    speaker_embedding.weight [num_speakers,512] is added to every encoder position."""
    emb = _t(sd, "speaker_embedding.weight", dtype)
    return enc + emb[torch.as_tensor(speaker_ids, dtype=torch.long)][:, None, :]


class T2DecoderOracle:
    """Decoder.inference / Decoder.decode, Tacotron2/TTS/models.py:578-673,
    with OriginalAttention.forward (:314-361) and Prenet (:76-82).

    `attn` selects the optional attention variants of the reference (all off in the repo config):
      attn_type   "original" | "graves" (GravesAttention, :106-183, K = attn_K mixtures)
      location_attn (:322-327, 269-273), windowing (:275-288), forward_attn / trans_agent /
      forward_attn_mask (:290-312, 357-360).
    The reference evaluates windowing with `.item()` on utterance 0 and is batch-1 only; here every
    utterance keeps its own window index, which is the same thing at batch 1."""

    ATTN_DEFAULT = dict(attn_type="original", location_attn=True, windowing=False, forward_attn=False,
                        trans_agent=False, forward_attn_mask=False, attn_K=5)

    def __init__(self, sd, r=7, r_init=7, attn_norm="sigmoid", prenet_type="bn",
                 dtype=torch.float32, attn=None):
        self.r, self.r_init, self.attn_norm, self.prenet_type = r, r_init, attn_norm, prenet_type
        self.dtype = dtype
        self.sd = sd
        self.attn = dict(self.ATTN_DEFAULT)
        if attn:
            self.attn.update(attn)
        g = lambda k: _t(sd, k, dtype)
        self.pre_w = [g("decoder.prenet.layers.%d.linear_layer.weight" % i) for i in range(2)]
        self.a_w_ih, self.a_w_hh = g("decoder.attention_rnn.weight_ih"), g("decoder.attention_rnn.weight_hh")
        self.a_b_ih, self.a_b_hh = g("decoder.attention_rnn.bias_ih"), g("decoder.attention_rnn.bias_hh")
        self.d_w_ih, self.d_w_hh = g("decoder.decoder_rnn.weight_ih"), g("decoder.decoder_rnn.weight_hh")
        self.d_b_ih, self.d_b_hh = g("decoder.decoder_rnn.bias_ih"), g("decoder.decoder_rnn.bias_hh")
        if self.attn["attn_type"] == "graves":
            self.g_w1, self.g_b1 = g("decoder.attention.N_a.0.weight"), g("decoder.attention.N_a.0.bias")
            self.g_w2, self.g_b2 = g("decoder.attention.N_a.2.weight"), g("decoder.attention.N_a.2.bias")
        else:
            self.w_q = g("decoder.attention.query_layer.linear_layer.weight")
            self.w_in = g("decoder.attention.inputs_layer.linear_layer.weight")
            self.w_v = g("decoder.attention.v.linear_layer.weight")
            self.b_v = g("decoder.attention.v.linear_layer.bias")
            if self.attn["location_attn"]:
                self.w_lc = g("decoder.attention.location_layer.location_conv.weight")
                self.w_ld = g("decoder.attention.location_layer.location_dense.linear_layer.weight")
            if self.attn["trans_agent"]:
                self.w_ta, self.b_ta = g("decoder.attention.ta.weight"), g("decoder.attention.ta.bias")
        self.w_o = g("decoder.linear_projection.linear_layer.weight")
        self.b_o = g("decoder.linear_projection.linear_layer.bias")
        self.w_s = g("decoder.stopnet.1.linear_layer.weight")
        self.b_s = g("decoder.stopnet.1.linear_layer.bias")

    def prenet(self, x):
        for i in range(2):
            x = x @ self.pre_w[i].t()
            if self.prenet_type == "bn":
                x = _bn_eval(x, self.sd, "decoder.prenet.layers.%d.bn" % i, self.dtype)
            x = torch.relu(x)
        return x

    def init(self, enc, enc_len=None, keep_states=False):
        """Decoder._init_states (:536-552) + attention.init_states (:240-249, 132-136).  With
        `keep_states` the recurrent state of the previous utterance survives (inference_truncated,
        :675-686); the attention state is reset either way."""
        B, T, _ = enc.shape
        z = lambda n: torch.zeros(B, n, dtype=self.dtype)
        self.enc = enc
        if not keep_states:
            self.h_a, self.c_a, self.h_d, self.c_d = z(1024), z(1024), z(1024), z(1024)
            self.ctx = z(512)
        self.a_prev, self.a_cum = z(T), z(T)
        self.lens = torch.full((B,), T, dtype=torch.long) if enc_len is None else torch.as_tensor(enc_len).long()
        self.mask = None
        if enc_len is not None:
            self.mask = torch.arange(T)[None, :] < self.lens[:, None]
        if self.attn["attn_type"] == "graves":
            self.mu_prev = z(self.attn["attn_K"])
            return
        self.P = enc @ self.w_in.t()                                     # :251-252
        if self.attn["forward_attn"]:                                    # :227-233
            self.alpha = torch.cat([torch.ones(B, 1, dtype=self.dtype),
                                    torch.zeros(B, T, dtype=self.dtype)[:, :-1] + 1e-7], dim=1)
            if self.mask is not None:
                self.alpha = self.alpha * self.mask
            self.u = 0.5 * torch.ones(B, 1, dtype=self.dtype)
        if self.attn["windowing"]:
            self.win_idx = [-1] * B                                      # :222-225

    # ---- attention variants ----------------------------------------------------------------
    def _graves(self):
        """GravesAttention.forward, :143-183."""
        K, T = self.attn["attn_K"], self.enc.shape[1]
        gbk = torch.relu(self.h_a @ self.g_w1.t() + self.g_b1) @ self.g_w2.t() + self.g_b2
        gbk = gbk.view(gbk.size(0), -1, K)
        g_t, b_t, k_t = gbk[:, 0, :], gbk[:, 1, :], gbk[:, 2, :]
        sig_t = F.softplus(b_t) + 1e-5
        mu_t = self.mu_prev + F.softplus(k_t)
        g_t = torch.softmax(g_t, dim=-1) + 1e-5
        j = (torch.arange(0, T + 1) + 0.5).to(self.dtype)
        phi = g_t.unsqueeze(-1) * (1 / (1 + torch.sigmoid((mu_t.unsqueeze(-1) - j) / sig_t.unsqueeze(-1))))
        cdf = phi.sum(1)
        a = cdf[:, 1:] - cdf[:, :-1]
        a = torch.where(a == 0, torch.full_like(a, 1e-8), a)
        if self.mask is not None:
            a = a.masked_fill(~self.mask, 1e-8)
        self.mu_prev = mu_t
        return a

    def _window(self, e):
        """apply_windowing, :275-288, per utterance (win_back 2, win_front 6)."""
        e = e.clone()
        for b in range(e.shape[0]):
            n = int(self.lens[b])
            back, front = self.win_idx[b] - 2, self.win_idx[b] + 6
            if back > 0:
                e[b, :back] = -float("inf")
            if front < n:
                e[b, front:] = -float("inf")
            if self.win_idx[b] == -1:
                e[b, 0] = e[b].max()
            self.win_idx[b] = int(torch.argmax(e[b]))
        return e

    def _forward_attention(self, alignment):
        """apply_forward_attention, :290-312.  The index arithmetic of the masking branch is
        Python's: n-1 and n-2 may be negative and then count from the end of the utterance."""
        shifted = F.pad(self.alpha[:, :-1].clone(), (1, 0, 0, 0))
        alpha = ((1 - self.u) * self.alpha + self.u * shifted + 1e-8) * alignment
        if self.attn["forward_attn_mask"]:
            for b in range(alignment.shape[0]):
                T = int(self.lens[b])
                n = int(torch.argmax(shifted[b, :T]))
                val = alpha[b, :T].max()
                row = alpha[b, :T]          # a view: the utterance's own positions
                row[n + 3:] = 0
                row[:(n - 1)] = 0
                row[n - 2] = 0.01 * val
        return alpha / alpha.sum(dim=1, keepdim=True)

    def _attention(self):
        if self.attn["attn_type"] == "graves":
            a = self._graves()
            self.a_prev = a
            self.ctx = torch.bmm(a.unsqueeze(1), self.enc).squeeze(1)
            return a
        q = self.h_a @ self.w_q.t()                                      # [B,128]
        if self.attn["location_attn"]:
            cat = torch.stack([self.a_prev, self.a_cum], dim=1)          # [B,2,T]
            loc = F.conv1d(cat, self.w_lc, padding=15).transpose(1, 2) @ self.w_ld.t()
            e = (torch.tanh(q[:, None, :] + loc + self.P) @ self.w_v.t()).squeeze(-1) + self.b_v
        else:
            e = (torch.tanh(q[:, None, :] + self.P) @ self.w_v.t()).squeeze(-1) + self.b_v
        if self.mask is not None:
            e = e.masked_fill(~self.mask, -float("inf"))
        if self.attn["windowing"]:
            e = self._window(e)
        if self.attn_norm == "softmax":
            a = torch.softmax(e, dim=-1)
        elif self.attn_norm == "sigmoid":
            s = torch.sigmoid(e)
            a = s / s.sum(dim=1, keepdim=True)
        else:
            raise ValueError("Unknown value for attention norm type")
        if self.attn["location_attn"]:
            self.a_cum = self.a_cum + a
        if self.attn["forward_attn"]:
            a = self._forward_attention(a)
            self.alpha = a
        self.a_prev = a
        self.ctx = torch.bmm(a.unsqueeze(1), self.enc).squeeze(1)
        if self.attn["forward_attn"] and self.attn["trans_agent"]:
            self.u = torch.sigmoid(torch.cat([self.ctx, self.h_a], dim=-1) @ self.w_ta.t() + self.b_ta)
        return a

    def step(self, memory):
        """One decode step; `memory` is the previous last mel frame [B,80]."""
        x2 = self.prenet(memory)
        self.h_a, self.c_a = lstm_cell(torch.cat([x2, self.ctx], 1), self.h_a, self.c_a,
                                       self.a_w_ih, self.a_w_hh, self.a_b_ih, self.a_b_hh)
        a = self._attention()
        self.h_d, self.c_d = lstm_cell(torch.cat([self.h_a, self.ctx], 1), self.h_d, self.c_d,
                                       self.d_w_ih, self.d_w_hh, self.d_b_ih, self.d_b_hh)
        o = torch.cat([self.h_d, self.ctx], 1) @ self.w_o.t() + self.b_o  # [B,80*r_init]
        s = torch.cat([self.h_d, o], 1) @ self.w_s.t() + self.b_s         # [B,1]
        return o[:, :80 * self.r], a, torch.sigmoid(s)

    def state(self):
        return dict(h_a=self.h_a, c_a=self.c_a, h_d=self.h_d, c_d=self.c_d, ctx=self.ctx,
                    a_prev=self.a_prev, a_cum=self.a_cum)

    def run(self, enc, max_steps=1000, stop_threshold=0.7, fixed_steps=False, enc_len=None,
            trace_steps=0, keep_states=False, memory=None):
        """Greedy loop.  For B > 1 every utterance keeps stepping until all have
        stopped (or `max_steps`); `n_steps[b]` records the first step whose
        stop token exceeded the threshold (the stop frame is kept), which is
        what a batch-1 run of that utterance returns.  `memory` (with `keep_states`) is the
        frame the first step is fed instead of the all-zero go frame; the frame the loop would
        feed next is left in `self.memory_next`."""
        B = enc.shape[0]
        self.init(enc, enc_len, keep_states)
        if memory is None:
            memory = torch.zeros(B, 80, dtype=self.dtype)
        outs, aligns, stops, trace = [], [], [], []
        n_steps = torch.full((B,), -1, dtype=torch.long)
        while True:
            o, a, s = self.step(memory)
            outs.append(o), aligns.append(a), stops.append(s)
            if len(trace) < trace_steps:
                trace.append({k: v.clone() for k, v in self.state().items()})
            t = len(outs)
            newly = (s[:, 0] > stop_threshold) & (n_steps < 0)
            n_steps[newly] = t
            if not fixed_steps and bool((n_steps > 0).all()):
                break
            if t == max_steps:
                break
            memory = o[:, 80 * (self.r - 1):80 * self.r]
        self.memory_next = memory
        n_steps[n_steps < 0] = len(outs)
        mel = torch.stack(outs, 1).reshape(B, -1, 80).transpose(1, 2)    # [B,80,F]
        return dict(mel=mel, align=torch.stack(aligns, 1), stop=torch.stack(stops, 1),
                    n_steps=n_steps, trace=trace)


def postnet(sd, mel, dtype=torch.float32):
    """Postnet.forward + residual, Tacotron2/TTS/models.py:414-417, 809-810."""
    x = mel
    for i in range(5):
        x = conv_bn_block(x, sd, "postnet.convolutions.%d" % i, "tanh" if i < 4 else None, dtype)
    return mel + x


def inference(sd, ids, r=7, r_init=7, attn_norm="sigmoid", prenet_type="bn", max_steps=1000,
              fixed_steps=False, dtype=torch.float32, trace_steps=0, enc_len=None, attn=None,
              speaker_ids=None):
    """Tacotron2.inference, Tacotron2/TTS/models.py:802-813.  Returns a dict with
    decoder_output [B,F,80], postnet_output [B,F,80], alignments [B,T_dec,T],
    stop_tokens [B,T_dec,1] (post-sigmoid), n_steps [B], encoder_outputs."""
    with torch.no_grad():
        enc = encoder(sd, ids, dtype)
        if speaker_ids is not None:
            enc = add_speaker_embedding(sd, enc, speaker_ids, dtype)
        dec = T2DecoderOracle(sd, r, r_init, attn_norm, prenet_type, dtype, attn)
        out = dec.run(enc, max_steps=max_steps, fixed_steps=fixed_steps, trace_steps=trace_steps,
                      enc_len=enc_len)
        post = postnet(sd, out["mel"], dtype)
    return dict(decoder_output=out["mel"].transpose(1, 2), postnet_output=post.transpose(1, 2),
                alignments=out["align"], stop_tokens=out["stop"], n_steps=out["n_steps"],
                encoder_outputs=enc, trace=out["trace"])


class TruncatedOracle:
    """Tacotron2.inference_truncated, Tacotron2/TTS/models.py:815-829 with
    Encoder.inference_truncated (:456-464) and Decoder.inference_truncated (:675-709): the encoder
    BiLSTM state, the decoder's recurrent state and the fed-back frame survive from one call to
    the next; the attention state is reset per call.  The reference feeds the whole decoder output
    back (`memory_truncated = decoder_output`, :703), which only fits the prenet at r = 1."""

    def __init__(self, sd, attn_norm="sigmoid", prenet_type="bn", dtype=torch.float32, attn=None,
                 max_steps=1000):
        self.sd, self.dtype, self.max_steps = sd, dtype, max_steps
        self.dec = T2DecoderOracle(sd, 1, 7, attn_norm, prenet_type, dtype, attn)
        self.rnn_state = None
        self.memory = None

    def __call__(self, ids):
        with torch.no_grad():
            enc, self.rnn_state = encoder(self.sd, ids, self.dtype, self.rnn_state, return_state=True)
            first = self.memory is None
            out = self.dec.run(enc, max_steps=self.max_steps, keep_states=not first, memory=self.memory)
            # the frame fed next is the last emitted one, stop frame included (:703 runs only when
            # the loop continues, so after a stop the previous frame stays)
            n = out["mel"].shape[2]
            self.memory = out["mel"][:, :, n - 2] if n > 1 else (self.memory if not first else torch.zeros(1, 80, dtype=self.dtype))
            post = postnet(self.sd, out["mel"], self.dtype)
        return dict(decoder_output=out["mel"].transpose(1, 2), postnet_output=post.transpose(1, 2),
                    alignments=out["align"], stop_tokens=out["stop"])
