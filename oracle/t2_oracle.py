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


def encoder(sd, ids, dtype=torch.float32):
    """Embedding + Encoder.inference, Tacotron2/TTS/models.py:803-804, 449-454."""
    emb = _t(sd, "embedding.weight", dtype)
    x = emb[torch.as_tensor(ids, dtype=torch.long)].transpose(1, 2)      # [B,512,T]
    for i in range(3):
        x = conv_bn_block(x, sd, "encoder.convolutions.%d" % i, "relu", dtype)
    x = x.transpose(1, 2)                                                # [B,T,512]
    B, T, _ = x.shape
    outs = []
    for sfx, order in (("_l0", range(T)), ("_l0_reverse", range(T - 1, -1, -1))):
        w_ih = _t(sd, "encoder.lstm.weight_ih" + sfx, dtype)
        w_hh = _t(sd, "encoder.lstm.weight_hh" + sfx, dtype)
        b_ih = _t(sd, "encoder.lstm.bias_ih" + sfx, dtype)
        b_hh = _t(sd, "encoder.lstm.bias_hh" + sfx, dtype)
        h = torch.zeros(B, 256, dtype=dtype)
        c = torch.zeros(B, 256, dtype=dtype)
        seq = [None] * T
        for t in order:
            h, c = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            seq[t] = h
        outs.append(torch.stack(seq, dim=1))
    return torch.cat(outs, dim=2)                                        # [B,T,512]


class T2DecoderOracle:
    """Decoder.inference / Decoder.decode, Tacotron2/TTS/models.py:578-673,
    with OriginalAttention.forward (:314-361) and Prenet (:76-82)."""

    def __init__(self, sd, r=7, r_init=7, attn_norm="sigmoid", prenet_type="bn",
                 dtype=torch.float32):
        self.r, self.r_init, self.attn_norm, self.prenet_type = r, r_init, attn_norm, prenet_type
        self.dtype = dtype
        self.sd = sd
        g = lambda k: _t(sd, k, dtype)
        self.pre_w = [g("decoder.prenet.layers.%d.linear_layer.weight" % i) for i in range(2)]
        self.a_w_ih, self.a_w_hh = g("decoder.attention_rnn.weight_ih"), g("decoder.attention_rnn.weight_hh")
        self.a_b_ih, self.a_b_hh = g("decoder.attention_rnn.bias_ih"), g("decoder.attention_rnn.bias_hh")
        self.d_w_ih, self.d_w_hh = g("decoder.decoder_rnn.weight_ih"), g("decoder.decoder_rnn.weight_hh")
        self.d_b_ih, self.d_b_hh = g("decoder.decoder_rnn.bias_ih"), g("decoder.decoder_rnn.bias_hh")
        self.w_q = g("decoder.attention.query_layer.linear_layer.weight")
        self.w_in = g("decoder.attention.inputs_layer.linear_layer.weight")
        self.w_v = g("decoder.attention.v.linear_layer.weight")
        self.b_v = g("decoder.attention.v.linear_layer.bias")
        self.w_lc = g("decoder.attention.location_layer.location_conv.weight")
        self.w_ld = g("decoder.attention.location_layer.location_dense.linear_layer.weight")
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

    def init(self, enc, enc_len=None):
        B, T, _ = enc.shape
        z = lambda n: torch.zeros(B, n, dtype=self.dtype)
        self.enc = enc
        self.P = enc @ self.w_in.t()                                     # :251-252
        self.h_a, self.c_a, self.h_d, self.c_d = z(1024), z(1024), z(1024), z(1024)
        self.ctx, self.a_prev, self.a_cum = z(512), z(T), z(T)
        self.mask = None
        if enc_len is not None:
            self.mask = torch.arange(T)[None, :] < torch.as_tensor(enc_len)[:, None]

    def step(self, memory):
        """One decode step; `memory` is the previous last mel frame [B,80]."""
        x2 = self.prenet(memory)
        self.h_a, self.c_a = lstm_cell(torch.cat([x2, self.ctx], 1), self.h_a, self.c_a,
                                       self.a_w_ih, self.a_w_hh, self.a_b_ih, self.a_b_hh)
        q = self.h_a @ self.w_q.t()                                      # [B,128]
        cat = torch.stack([self.a_prev, self.a_cum], dim=1)              # [B,2,T]
        loc = F.conv1d(cat, self.w_lc, padding=15).transpose(1, 2) @ self.w_ld.t()
        e = (torch.tanh(q[:, None, :] + loc + self.P) @ self.w_v.t()).squeeze(-1) + self.b_v
        if self.mask is not None:
            e = e.masked_fill(~self.mask, -float("inf"))
        if self.attn_norm == "softmax":
            a = torch.softmax(e, dim=-1)
        elif self.attn_norm == "sigmoid":
            s = torch.sigmoid(e)
            a = s / s.sum(dim=1, keepdim=True)
        else:
            raise ValueError("Unknown value for attention norm type")
        self.a_cum = self.a_cum + a
        self.a_prev = a
        self.ctx = torch.bmm(a.unsqueeze(1), self.enc).squeeze(1)
        self.h_d, self.c_d = lstm_cell(torch.cat([self.h_a, self.ctx], 1), self.h_d, self.c_d,
                                       self.d_w_ih, self.d_w_hh, self.d_b_ih, self.d_b_hh)
        o = torch.cat([self.h_d, self.ctx], 1) @ self.w_o.t() + self.b_o  # [B,80*r_init]
        s = torch.cat([self.h_d, o], 1) @ self.w_s.t() + self.b_s         # [B,1]
        return o[:, :80 * self.r], a, torch.sigmoid(s)

    def state(self):
        return dict(h_a=self.h_a, c_a=self.c_a, h_d=self.h_d, c_d=self.c_d, ctx=self.ctx,
                    a_prev=self.a_prev, a_cum=self.a_cum)

    def run(self, enc, max_steps=1000, stop_threshold=0.7, fixed_steps=False, enc_len=None,
            trace_steps=0):
        """Greedy loop.  For B > 1 every utterance keeps stepping until all have
        stopped (or `max_steps`); `n_steps[b]` records the first step whose
        stop token exceeded the threshold (the stop frame is kept), which is
        what a batch-1 run of that utterance returns."""
        B = enc.shape[0]
        self.init(enc, enc_len)
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
              fixed_steps=False, dtype=torch.float32, trace_steps=0, enc_len=None):
    """Tacotron2.inference, Tacotron2/TTS/models.py:802-813.  Returns a dict with
    decoder_output [B,F,80], postnet_output [B,F,80], alignments [B,T_dec,T],
    stop_tokens [B,T_dec,1] (post-sigmoid), n_steps [B], encoder_outputs."""
    with torch.no_grad():
        enc = encoder(sd, ids, dtype)
        dec = T2DecoderOracle(sd, r, r_init, attn_norm, prenet_type, dtype)
        out = dec.run(enc, max_steps=max_steps, fixed_steps=fixed_steps, trace_steps=trace_steps,
                      enc_len=enc_len)
        post = postnet(sd, out["mel"], dtype)
    return dict(decoder_output=out["mel"].transpose(1, 2), postnet_output=post.transpose(1, 2),
                alignments=out["align"], stop_tokens=out["stop"], n_steps=out["n_steps"],
                encoder_outputs=enc, trace=out["trace"])
