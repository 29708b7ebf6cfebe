"""CPU oracle for the Tacotron (v1) model (TEST INFRASTRUCTURE ONLY).

Restates, in torch CPU tensor primitives, `Tacotron.forward` in eval mode
(Tacotron/TTS/models.py:692-700): embedding -> Encoder (Prenet :250-253, CBHG
:412-434) -> greedy Decoder.forward (:558-645, AttentionRNNCell :124-156,
LocationSensitiveAttention :71-88, StopNet :663-667) -> PostCBHG -> last_linear
+ sigmoid.  Pinned by `tests/golden/t1_*.npz`, which `oracle/gen_golden.py t1`
produced by running the reference itself.  Nothing in `tts_amd/` imports this.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, key, dtype=torch.float32):
    return torch.as_tensor(np.asarray(sd[key])).to(dtype)


def _linear(sd, prefix, x, bias=True):
    w = _t(sd, prefix + ".weight")
    return x @ w.t() + (_t(sd, prefix + ".bias") if bias else 0.0)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRUCell math, gate order r, z, n."""
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def bn_conv1d(sd, prefix, x, k, act):
    """BatchNormConv1d.forward (:307-313): pad [(k-1)//2, k//2], conv (no bias), BN eps 1e-3, act."""
    w = _t(sd, prefix + ".conv1d.weight")
    x = F.pad(x, ((k - 1) // 2, k // 2))
    x = F.conv1d(x, w)
    g, b = _t(sd, prefix + ".bn.weight"), _t(sd, prefix + ".bn.bias")
    m, v = _t(sd, prefix + ".bn.running_mean"), _t(sd, prefix + ".bn.running_var")
    x = (x - m[None, :, None]) / torch.sqrt(v[None, :, None] + 1e-3) * g[None, :, None] + b[None, :, None]
    return torch.relu(x) if act else x


def cbhg(sd, prefix, inputs, K, n_proj, pre_highway):
    """CBHG.forward (:412-434) on inputs [B,T,C]."""
    x = inputs.transpose(1, 2)
    outs = [bn_conv1d(sd, "%s.conv1d_banks.%d" % (prefix, k - 1), x, k, True) for k in range(1, K + 1)]
    x = torch.cat(outs, dim=1)
    x = F.max_pool1d(F.pad(x, (0, 1)), kernel_size=2, stride=1)
    for i in range(n_proj):
        x = bn_conv1d(sd, "%s.conv1d_projections.%d" % (prefix, i), x, 3, i < n_proj - 1)
    x = x.transpose(1, 2) + inputs
    if pre_highway:
        x = _linear(sd, prefix + ".pre_highway", x, bias=False)
    for i in range(4):
        H = torch.relu(_linear(sd, "%s.highways.%d.H" % (prefix, i), x))
        Tg = torch.sigmoid(_linear(sd, "%s.highways.%d.T" % (prefix, i), x))
        x = H * Tg + x * (1.0 - Tg)
    B, T, _ = x.shape
    dirs = []
    for sfx, order in (("_l0", range(T)), ("_l0_reverse", range(T - 1, -1, -1))):
        w_ih, w_hh = _t(sd, prefix + ".gru.weight_ih" + sfx), _t(sd, prefix + ".gru.weight_hh" + sfx)
        b_ih, b_hh = _t(sd, prefix + ".gru.bias_ih" + sfx), _t(sd, prefix + ".gru.bias_hh" + sfx)
        h = torch.zeros(B, w_hh.shape[1])
        seq = [None] * T
        for t in order:
            h = gru_cell(x[:, t], h, w_ih, w_hh, b_ih, b_hh)
            seq[t] = h
        dirs.append(torch.stack(seq, 1))
    return torch.cat(dirs, 2)


def encoder(sd, ids):
    x = _t(sd, "embedding.weight")[torch.as_tensor(ids, dtype=torch.long)]
    for i in range(2):
        x = torch.relu(_linear(sd, "encoder.prenet.layers.%d" % i, x))
    return cbhg(sd, "encoder.cbhg.cbhg", x, 16, 2, False)


def decoder(sd, enc, r=2, memory_size=5, max_decoder_steps=500):
    """Greedy Decoder.forward (:586-645), batch 1 (the stop rule uses .item())."""
    B, T, _ = enc.shape
    mem = _t(sd, "decoder.memory_init.weight")[0:1].repeat(B, 1)
    h_att = _t(sd, "decoder.attention_rnn_init.weight")[0:1].repeat(B, 1)
    h_dec = [_t(sd, "decoder.decoder_rnn_inits.weight")[i:i + 1].repeat(B, 1) for i in range(2)]
    ctx = torch.zeros(B, enc.shape[2])
    att, att_cum = torch.zeros(B, T), torch.zeros(B, T)
    am = "decoder.attention_rnn.alignment_model"
    annot = _linear(sd, am + ".annot_layer", enc)
    g = lambda p: [_t(sd, p + s) for s in (".weight_ih", ".weight_hh", ".bias_ih", ".bias_hh")]
    att_rnn, dec_rnn = g("decoder.attention_rnn.rnn_cell"), [g("decoder.decoder_rnns.%d" % i) for i in range(2)]
    outputs, attentions, stops = [], [], []
    t = 0
    while True:
        if t > 0:
            mem = torch.cat([mem[:, r * 80:], outputs[-1]], dim=-1)
        p = mem
        for i in range(2):
            p = torch.relu(_linear(sd, "decoder.prenet.layers.%d" % i, p))
        h_att = gru_cell(torch.cat([p, ctx], -1), h_att, *att_rnn)
        loc = F.conv1d(F.pad(torch.stack([att, att_cum], 1), (15, 15)), _t(sd, am + ".loc_conv.1.weight"))
        ploc = _linear(sd, am + ".loc_linear", loc.transpose(1, 2))
        pq = _linear(sd, am + ".query_layer", h_att.unsqueeze(1))
        e = (torch.tanh(pq + annot + ploc) @ _t(sd, am + ".v.weight").t()).squeeze(-1)
        s = torch.sigmoid(e)
        att = s / s.sum(dim=1, keepdim=True)
        ctx = torch.bmm(att.unsqueeze(1), enc).squeeze(1)
        att_cum = att_cum + att
        d = _linear(sd, "decoder.project_to_decoder_in", torch.cat([h_att, ctx], -1))
        for i in range(2):
            h_dec[i] = gru_cell(d, h_dec[i], *dec_rnn[i])
            d = h_dec[i] + d
        out = torch.sigmoid(_linear(sd, "decoder.proj_to_mel", d))
        stop = torch.sigmoid(_linear(sd, "decoder.stopnet.linear", torch.cat([d, out], -1)))
        outputs.append(out), attentions.append(att), stops.append(stop)
        t += 1
        if t > T / 4 and (float(stop[0, 0]) > 0.6 or float(att[0, -1]) > 0.6):
            break
        if t > max_decoder_steps:
            break
    return torch.stack(outputs, 1), torch.stack(attentions, 1), torch.stack(stops, 1)


def postnet(sd, mel):
    """PostCBHG + last_linear + sigmoid (:698-699) on mel [B,F,80]."""
    x = cbhg(sd, "postnet.cbhg", mel, 8, 2, True)
    return torch.sigmoid(_linear(sd, "last_linear.0", x))


def forward(sd, ids, r=2, memory_size=5, max_decoder_steps=500):
    with torch.no_grad():
        enc = encoder(sd, ids)
        out, att, stop = decoder(sd, enc, r, memory_size, max_decoder_steps)
        B = out.shape[0]
        mel = out.reshape(B, -1, 80)
        lin = postnet(sd, mel)
    return dict(mel=mel, linear=lin, alignments=att, stop_tokens=stop, encoder_outputs=enc)
