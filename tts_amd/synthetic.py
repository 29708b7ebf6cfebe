"""Deterministic synthetic checkpoints and inputs.

The trained checkpoints of the reference are Git-LFS pointer stubs
(/root/reference/.gitattributes:1-2), so every test and benchmark runs on a
synthetic weight set laid out exactly like the reference `state_dict`
(keys/shapes of Tacotron2/TTS/models.py:727-771 and
Tacotron/TTS/models.py:669-690).  The generator is a counter-based hash
(splitmix64) evaluated with NumPy integer arithmetic, so the same seed gives
the same bits on any NumPy / torch version and on any machine: golden vectors
store only the seed, not 116 MB of weights.

Default-initialised weights give a degenerate decoder (uniform alignments,
stop token stuck at 0.50); the gains below were chosen so that mel frames have
O(1) magnitude, the stop token crosses 0.7 at a data-dependent step, the
alignments move, and an fp32 and an fp64 evaluation of the same free-running
decode still agree to ~1e-4 after 400 steps (larger recurrent gains make the
decoder chaotic: rounding differences then grow past any tolerance).
"""
import zlib
from collections import OrderedDict

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _stream_key(seed, name):
    h = np.uint64(zlib.crc32(name.encode("utf8")))
    k = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001) + h],
                             dtype=np.uint64))[0]
    return k


def uniform01(seed, name, n):
    """n doubles in [0, 1), a pure function of (seed, name, index)."""
    key = _stream_key(seed, name)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, name, shape, lo, hi):
    n = int(np.prod(shape))
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, name, shape, mean, std):
    n = int(np.prod(shape))
    u1 = uniform01(seed, name + "/u1", n)
    u2 = uniform01(seed, name + "/u2", n)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def integers(seed, name, shape, lo, hi):
    """Integers in [lo, hi)."""
    n = int(np.prod(shape))
    u = uniform01(seed, name, n)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64).reshape(shape)


# --------------------------------------------------------------------------
# Tacotron2
# --------------------------------------------------------------------------
T2_GAINS = dict(lstm=2.5, proj=6.0, v=10.0, loc=10.0, query=3.0, stop=8.0,
                inputs=1.0, prenet=1.0, conv=1.0, emb=3.0, enc_lstm=4.0,
                stop_bias=-7.8, stop_stream=None)

# Gains of a checkpoint whose stop token fires late: with the default stopnet weights the stop logit peaks
# during the first 15 steps of any text and never comes back, so a free run ends there.  Stopnet weights
# drawn from hash stream 18 give a logit that rises slowly to a hump around step 355 of the 100-id
# utterance t2_text_ids(99, 1, 100); with this bias it crosses 0.7 near step 309, on a slope of 0.01 per
# step, so the stop step is data-dependent and robust to rounding.
T2_LATE_STOP_GAINS = dict(stop_stream=18, stop_bias=-20.65)


def _xavier(seed, name, shape, gain=1.0):
    fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
    a = gain * np.sqrt(6.0 / (fan_in + fan_out))
    return uniform(seed, name, shape, -a, a)


def _bn(sd, seed, prefix, n):
    sd[prefix + ".weight"] = uniform(seed, prefix + ".weight", (n,), 0.5, 1.5)
    sd[prefix + ".bias"] = normal(seed, prefix + ".bias", (n,), 0.0, 0.1)
    sd[prefix + ".running_mean"] = normal(seed, prefix + ".running_mean", (n,), 0.0, 0.1)
    sd[prefix + ".running_var"] = uniform(seed, prefix + ".running_var", (n,), 0.5, 1.5)
    sd[prefix + ".num_batches_tracked"] = np.array(1000, dtype=np.int64)


def _conv_bn(sd, seed, prefix, cout, cin, k, gain=1.0):
    a = gain / np.sqrt(cin * k)
    sd[prefix + ".net.0.weight"] = uniform(seed, prefix + ".net.0.weight", (cout, cin, k), -a, a)
    sd[prefix + ".net.0.bias"] = uniform(seed, prefix + ".net.0.bias", (cout,), -a, a)
    _bn(sd, seed, prefix + ".net.1", cout)


def _lstm(sd, seed, prefix, in_dim, hid, gain, suffix=""):
    a = gain / np.sqrt(hid)
    for nm, shape in (("weight_ih", (4 * hid, in_dim)), ("weight_hh", (4 * hid, hid)),
                      ("bias_ih", (4 * hid,)), ("bias_hh", (4 * hid,))):
        key = "%s.%s%s" % (prefix, nm, suffix)
        sd[key] = uniform(seed, key, shape, -a, a)


def t2_state_dict(seed=0, num_chars=129, r_init=7, prenet_type="bn", gains=None, attn_type="original",
                  location_attn=True, trans_agent=False, attn_K=5, num_speakers=0):
    """Synthetic Tacotron2 `state_dict` (NumPy arrays, reference key names).

    Shapes follow the reference constructor (Tacotron2/TTS/models.py:755-771,
    489-524, 405-412, 422-431).  The optional arguments add or drop the tensors of the
    attention variants exactly as the reference constructor does (:113-130 Graves N_a,
    :204-212 transition agent / location layer, :759-761 speaker embedding); the default
    arguments give the same bits as before those options existed."""
    g = dict(T2_GAINS)
    if gains:
        g.update(gains)
    sd = OrderedDict()
    val = np.sqrt(3.0) * np.sqrt(2.0 / (num_chars + 512))
    sd["embedding.weight"] = uniform(seed, "embedding.weight", (num_chars, 512),
                                     -val * g["emb"], val * g["emb"])
    for i in range(3):
        _conv_bn(sd, seed, "encoder.convolutions.%d" % i, 512, 512, 5, g["conv"])
    for sfx in ("_l0", "_l0_reverse"):
        _lstm(sd, seed, "encoder.lstm", 512, 256, g["enc_lstm"], sfx)
    # decoder
    dims = [80, 256]
    for i in range(2):
        p = "decoder.prenet.layers.%d" % i
        sd[p + ".linear_layer.weight"] = _xavier(seed, p + ".w", (256, dims[i]), g["prenet"])
        if prenet_type == "bn":
            _bn(sd, seed, p + ".bn", 256)
    _lstm(sd, seed, "decoder.attention_rnn", 768, 1024, g["lstm"])
    if attn_type == "graves":
        a = 1.0 / np.sqrt(1024.0)
        sd["decoder.attention.N_a.0.weight"] = uniform(seed, "att.N_a.0.w", (1024, 1024), -3 * a, 3 * a)
        sd["decoder.attention.N_a.0.bias"] = uniform(seed, "att.N_a.0.b", (1024,), -a, a)
        sd["decoder.attention.N_a.2.weight"] = uniform(seed, "att.N_a.2.w", (3 * attn_K, 1024), -a, a)
        b2 = uniform(seed, "att.N_a.2.b", (3 * attn_K,), -a, a)
        # GravesAttention.init_layers (:128-130) starts the std bias at 10 (alignments nearly flat
        # over a short text); 0.5 here so that the mixture is narrow enough to test the arithmetic
        b2[attn_K:2 * attn_K] = 0.5
        b2[2 * attn_K:] = 1.0             # bias mean
        sd["decoder.attention.N_a.2.bias"] = b2
    else:
        sd["decoder.attention.query_layer.linear_layer.weight"] = _xavier(
            seed, "att.query", (128, 1024), g["query"] * 5.0 / 3.0)
        sd["decoder.attention.inputs_layer.linear_layer.weight"] = _xavier(
            seed, "att.inputs", (128, 512), g["inputs"] * 5.0 / 3.0)
        sd["decoder.attention.v.linear_layer.weight"] = _xavier(seed, "att.v", (1, 128), g["v"])
        sd["decoder.attention.v.linear_layer.bias"] = uniform(seed, "att.v.b", (1,), -0.1, 0.1)
        if trans_agent:
            a = 4.0 / np.sqrt(1536.0)
            sd["decoder.attention.ta.weight"] = uniform(seed, "att.ta.w", (1, 1536), -a, a)
            sd["decoder.attention.ta.bias"] = uniform(seed, "att.ta.b", (1,), -0.5, 0.5)
        if location_attn:
            a = g["loc"] / np.sqrt(2 * 31)
            sd["decoder.attention.location_layer.location_conv.weight"] = uniform(
                seed, "att.loc_conv", (32, 2, 31), -a, a)
            sd["decoder.attention.location_layer.location_dense.linear_layer.weight"] = _xavier(
                seed, "att.loc_dense", (128, 32), 5.0 / 3.0)
    _lstm(sd, seed, "decoder.decoder_rnn", 1536, 1024, g["lstm"])
    sd["decoder.linear_projection.linear_layer.weight"] = _xavier(
        seed, "dec.proj", (80 * r_init, 1536), g["proj"])
    sd["decoder.linear_projection.linear_layer.bias"] = uniform(
        seed, "dec.proj.b", (80 * r_init,), -0.05, 0.05)
    stop_name = "dec.stop" if g["stop_stream"] is None else "dec.stop/%d" % g["stop_stream"]
    sd["decoder.stopnet.1.linear_layer.weight"] = _xavier(
        seed, stop_name, (1, 1024 + 80 * r_init), g["stop"])
    sd["decoder.stopnet.1.linear_layer.bias"] = np.full((1,), g["stop_bias"], dtype=np.float32)
    # postnet
    chans = [80, 512, 512, 512, 512, 80]
    for i in range(5):
        _conv_bn(sd, seed, "postnet.convolutions.%d" % i, chans[i + 1], chans[i], 5, g["conv"])
    if num_speakers > 1:
        sd["speaker_embedding.weight"] = normal(seed, "speaker_embedding.weight", (num_speakers, 512), 0.0, 0.3)
    return sd


def t2_text_ids(seed, n_utt, t_in, num_chars=129):
    """Synthetic phoneme IDs in [3, num_chars): ids 0-2 are pad/eos/bos
    (Tacotron2/TTS/utils.py:431-432)."""
    return integers(seed, "t2.ids", (n_utt, t_in), 3, num_chars)


def gl_init_phase(seed, n_freq, n_frames):
    """Explicit Griffin-Lim initial phase in [0, 1) (the reference draws it from
    the unseeded global NumPy RNG, Tacotron2/TTS/utils.py:923)."""
    return uniform01(seed, "gl.phase", n_freq * n_frames).reshape(n_freq, n_frames)


# --------------------------------------------------------------------------
# Tacotron (v1)
# --------------------------------------------------------------------------
def _lin(sd, seed, prefix, out_dim, in_dim, gain=1.0, bias=True):
    a = gain / np.sqrt(in_dim)
    sd[prefix + ".weight"] = uniform(seed, prefix + ".weight", (out_dim, in_dim), -a, a)
    if bias:
        sd[prefix + ".bias"] = uniform(seed, prefix + ".bias", (out_dim,), -a, a)


def _gru(sd, seed, prefix, in_dim, hid, gain=1.0, cell=True, suffix=""):
    a = gain / np.sqrt(hid)
    for nm, shape in (("weight_ih", (3 * hid, in_dim)), ("weight_hh", (3 * hid, hid)),
                      ("bias_ih", (3 * hid,)), ("bias_hh", (3 * hid,))):
        key = "%s.%s%s" % (prefix, nm, suffix)
        sd[key] = uniform(seed, key, shape, -a, a)


def _cbhg(sd, seed, prefix, in_dim, K, bank, projections, highway_dim, gru_dim, pre_highway, gain=1.0, hw=1.0):
    for k in range(1, K + 1):
        p = "%s.conv1d_banks.%d" % (prefix, k - 1)
        a = gain / np.sqrt(in_dim * k)
        sd[p + ".conv1d.weight"] = uniform(seed, p + ".conv1d.weight", (bank, in_dim, k), -a, a)
        _bn(sd, seed, p + ".bn", bank)
    cin = bank * K
    for i, cout in enumerate(projections):
        p = "%s.conv1d_projections.%d" % (prefix, i)
        a = gain / np.sqrt(cin * 3)
        sd[p + ".conv1d.weight"] = uniform(seed, p + ".conv1d.weight", (cout, cin, 3), -a, a)
        _bn(sd, seed, p + ".bn", cout)
        cin = cout
    if pre_highway:
        _lin(sd, seed, prefix + ".pre_highway", highway_dim, projections[-1], hw, bias=False)
    for i in range(4):
        _lin(sd, seed, "%s.highways.%d.H" % (prefix, i), highway_dim, highway_dim, hw)
        _lin(sd, seed, "%s.highways.%d.T" % (prefix, i), highway_dim, highway_dim, hw)
    for sfx in ("_l0", "_l0_reverse"):
        _gru(sd, seed, prefix + ".gru", highway_dim, gru_dim, hw, suffix=sfx)


def t1_state_dict(seed=0, num_chars=61, r=2, memory_size=5, gains=None):
    """Synthetic Tacotron (v1) `state_dict`; shapes follow
    Tacotron/TTS/models.py:336-410 (CBHG), 487-536 (Decoder), 669-690."""
    g = dict(rnn=3.0, proj=8.0, v=20.0, loc=6.0, stop=30.0, stop_bias=None, cbhg=2.0, last=8.0, hw=2.0)
    if gains:
        g.update(gains)
    sd = OrderedDict()
    sd["embedding.weight"] = normal(seed, "embedding.weight", (num_chars, 256), 0.0, 0.3)
    _lin(sd, seed, "encoder.prenet.layers.0", 256, 256)
    _lin(sd, seed, "encoder.prenet.layers.1", 128, 256)
    _cbhg(sd, seed, "encoder.cbhg.cbhg", 128, 16, 128, [128, 128], 128, 128, False, g["cbhg"], g["hw"])
    mem_in = 80 * memory_size if memory_size > 0 else 80 * r
    _lin(sd, seed, "decoder.prenet.layers.0", 256, mem_in)
    _lin(sd, seed, "decoder.prenet.layers.1", 128, 256)
    _gru(sd, seed, "decoder.attention_rnn.rnn_cell", 384, 256, g["rnn"])
    am = "decoder.attention_rnn.alignment_model"
    a = g["loc"] / np.sqrt(2 * 31)
    sd[am + ".loc_conv.1.weight"] = uniform(seed, am + ".loc_conv", (32, 2, 31), -a, a)
    _lin(sd, seed, am + ".loc_linear", 128, 32)
    _lin(sd, seed, am + ".query_layer", 128, 256, 2.0)
    _lin(sd, seed, am + ".annot_layer", 128, 256)
    _lin(sd, seed, am + ".v", 1, 128, g["v"], bias=False)
    _lin(sd, seed, "decoder.project_to_decoder_in", 256, 512)
    for i in range(2):
        _gru(sd, seed, "decoder.decoder_rnns.%d" % i, 256, 256, g["rnn"])
    _lin(sd, seed, "decoder.proj_to_mel", 80 * r, 256, g["proj"])
    sd["decoder.attention_rnn_init.weight"] = normal(seed, "dec.att_init", (1, 256), 0.0, 0.3)
    sd["decoder.memory_init.weight"] = normal(seed, "dec.mem_init", (1, mem_in), 0.0, 0.3)
    sd["decoder.decoder_rnn_inits.weight"] = normal(seed, "dec.rnn_inits", (2, 256), 0.0, 0.3)
    _lin(sd, seed, "decoder.stopnet.linear", 1, 256 + 80 * r, g["stop"])
    if g["stop_bias"] is not None:
        sd["decoder.stopnet.linear.bias"] = np.full((1,), g["stop_bias"], dtype=np.float32)
    _cbhg(sd, seed, "postnet.cbhg", 80, 8, 128, [256, 80], 128, 128, True, g["cbhg"], g["hw"])
    _lin(sd, seed, "last_linear.0", 1025, 256, g["last"])
    return sd


def t1_text_ids(seed, t_in, num_chars=61):
    return integers(seed, "t1.ids", (1, t_in), 2, num_chars)
