"""ctypes binding of libttsx.so (the C ABI declared in include/ttsx.h).

There is no CPU fallback: if the library is missing or a call fails, the
caller gets an exception.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("TTSX_LIB", "libttsx.so"))

OK, ERR_ARG, ERR_HIP, ERR_STATE = 0, -1, -2, -3
NORM_SOFTMAX, NORM_SIGMOID = 0, 1
PAD_CONSTANT, PAD_REFLECT = 0, 1

_fp = c_void_p  # device pointers travel as integers


class T2DecoderWeights(Structure):
    _fields_ = [
        ("prenet_w", _fp * 2), ("prenet_bn_weight", _fp * 2), ("prenet_bn_bias", _fp * 2),
        ("prenet_bn_mean", _fp * 2), ("prenet_bn_var", _fp * 2),
        ("attn_rnn_w_ih", _fp), ("attn_rnn_w_hh", _fp), ("attn_rnn_b_ih", _fp), ("attn_rnn_b_hh", _fp),
        ("query_w", _fp), ("inputs_w", _fp), ("v_w", _fp), ("v_b", _fp),
        ("loc_conv_w", _fp), ("loc_dense_w", _fp),
        ("dec_rnn_w_ih", _fp), ("dec_rnn_w_hh", _fp), ("dec_rnn_b_ih", _fp), ("dec_rnn_b_hh", _fp),
        ("proj_w", _fp), ("proj_b", _fp), ("stop_w", _fp), ("stop_b", _fp),
    ]


class T2AttentionOptions(Structure):
    _fields_ = [("windowing", c_int), ("forward_attn", c_int), ("forward_attn_mask", c_int), ("trans_agent", c_int),
                ("ta_w", _fp), ("ta_b", _fp), ("graves_K", c_int),
                ("graves_w1", _fp), ("graves_b1", _fp), ("graves_w2", _fp), ("graves_b2", _fp)]


class PostnetWeights(Structure):
    _fields_ = [("conv_w", _fp * 5), ("conv_b", _fp * 5), ("bn_weight", _fp * 5), ("bn_bias", _fp * 5),
                ("bn_mean", _fp * 5), ("bn_var", _fp * 5)]


class EncoderConvWeights(Structure):
    _fields_ = [("conv_w", _fp * 3), ("conv_b", _fp * 3), ("bn_weight", _fp * 3), ("bn_bias", _fp * 3),
                ("bn_mean", _fp * 3), ("bn_var", _fp * 3)]


class CbhgWeights(Structure):
    _fields_ = [("K", c_int), ("in_features", c_int), ("proj_dims", c_int * 2),
                ("bank_w", _fp * 16), ("bank_bn_weight", _fp * 16), ("bank_bn_bias", _fp * 16),
                ("bank_bn_mean", _fp * 16), ("bank_bn_var", _fp * 16),
                ("proj_w", _fp * 2), ("proj_bn_weight", _fp * 2), ("proj_bn_bias", _fp * 2),
                ("proj_bn_mean", _fp * 2), ("proj_bn_var", _fp * 2),
                ("pre_highway_w", _fp),
                ("hw_H_w", _fp * 4), ("hw_H_b", _fp * 4), ("hw_T_w", _fp * 4), ("hw_T_b", _fp * 4),
                ("gru_w_ih", _fp * 2), ("gru_w_hh", _fp * 2), ("gru_b_ih", _fp * 2), ("gru_b_hh", _fp * 2)]


class T1Weights(Structure):
    _fields_ = [("enc_prenet_w", _fp * 2), ("enc_prenet_b", _fp * 2),
                ("enc_cbhg", CbhgWeights),
                ("dec_prenet_w", _fp * 2), ("dec_prenet_b", _fp * 2),
                ("att_rnn_w_ih", _fp), ("att_rnn_w_hh", _fp), ("att_rnn_b_ih", _fp), ("att_rnn_b_hh", _fp),
                ("loc_conv_w", _fp), ("loc_linear_w", _fp), ("loc_linear_b", _fp),
                ("query_w", _fp), ("query_b", _fp), ("annot_w", _fp), ("annot_b", _fp), ("v_w", _fp),
                ("proj_dec_in_w", _fp), ("proj_dec_in_b", _fp),
                ("dec_rnn_w_ih", _fp * 2), ("dec_rnn_w_hh", _fp * 2), ("dec_rnn_b_ih", _fp * 2),
                ("dec_rnn_b_hh", _fp * 2),
                ("proj_mel_w", _fp), ("proj_mel_b", _fp),
                ("att_rnn_init", _fp), ("memory_init", _fp), ("dec_rnn_inits", _fp),
                ("stop_w", _fp), ("stop_b", _fp),
                ("post_cbhg", CbhgWeights),
                ("last_linear_w", _fp), ("last_linear_b", _fp)]


class EncoderLstmWeights(Structure):
    _fields_ = [("w_ih", _fp * 2), ("w_hh", _fp * 2), ("b_ih", _fp * 2), ("b_hh", _fp * 2)]


class StftParams(Structure):
    _fields_ = [("n_fft", c_int), ("hop_length", c_int), ("win_length", c_int), ("pad_mode", c_int)]


class AudioParams(Structure):
    _fields_ = [("signal_norm", c_int), ("symmetric_norm", c_int), ("clip_norm", c_int),
                ("max_norm", c_float), ("min_level_db", c_float), ("ref_level_db", c_float),
                ("power", c_float)]


# name -> (restype, argtypes); every symbol include/ttsx.h declares
PROTOTYPES = {
    "ttsx_last_error_string": (c_char_p, []),
    "ttsx_version": (c_int, []),
    "ttsx_t2_decoder_create": (c_int, [POINTER(T2DecoderWeights), c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "ttsx_t2_decoder_destroy": (c_int, [c_void_p]),
    "ttsx_t2_decoder_set_attention": (c_int, [c_void_p, POINTER(T2AttentionOptions), c_void_p]),
    "ttsx_t2_decoder_set_initial": (c_int, [c_void_p, c_int, _fp, c_int, c_void_p]),
    "ttsx_t2_decoder_run": (c_int, [c_void_p, _fp, _fp, c_int, c_int, c_int, c_int, c_float, c_int, _fp, _fp,
                                    _fp, _fp, POINTER(c_int), c_void_p]),
    "ttsx_t2_decoder_steps_launched": (c_int, [c_void_p]),
    "ttsx_t2_decoder_set_plan": (c_int, [c_void_p, c_int, c_int, c_int]),
    "ttsx_t2_decoder_set_team": (c_int, [c_void_p, c_int]),
    "ttsx_t2_decoder_set_fusion": (c_int, [c_void_p, c_int]),
    "ttsx_t2_decoder_set_pieces": (c_int, [c_void_p, c_int]),
    "ttsx_t2_decoder_set_polling": (c_int, [c_void_p, c_int, c_int]),
    "ttsx_t2_decoder_begin": (c_int, [c_void_p, _fp, _fp, c_int, c_int, c_void_p]),
    "ttsx_t2_decoder_load_state": (c_int, [c_void_p] + [_fp] * 7 + [c_void_p]),
    "ttsx_t2_decoder_step": (c_int, [c_void_p, _fp, c_int, _fp, _fp, _fp, c_void_p]),
    "ttsx_t2_decoder_read_state": (c_int, [c_void_p] + [_fp] * 7 + [c_void_p]),
    "ttsx_postnet_create": (c_int, [POINTER(PostnetWeights), c_void_p, POINTER(c_void_p)]),
    "ttsx_postnet_destroy": (c_int, [c_void_p]),
    "ttsx_postnet_run": (c_int, [c_void_p, _fp, _fp, c_int, c_int, _fp, c_void_p]),
    "ttsx_encoder_convs_create": (c_int, [POINTER(EncoderConvWeights), c_void_p, POINTER(c_void_p)]),
    "ttsx_encoder_convs_destroy": (c_int, [c_void_p]),
    "ttsx_encoder_convs_run": (c_int, [c_void_p, _fp, _fp, c_int, c_int, _fp, c_void_p]),
    "ttsx_encoder_lstm_create": (c_int, [POINTER(EncoderLstmWeights), c_void_p, POINTER(c_void_p)]),
    "ttsx_encoder_lstm_destroy": (c_int, [c_void_p]),
    "ttsx_encoder_lstm_run": (c_int, [c_void_p, _fp, _fp, c_int, c_int, _fp, c_void_p]),
    "ttsx_encoder_lstm_run_stateful": (c_int, [c_void_p, _fp, _fp, c_int, c_int, _fp, _fp, _fp, c_void_p]),
    "ttsx_encoder_lstm_set_plan": (c_int, [c_void_p, c_int]),
    "ttsx_add_speaker_embedding": (c_int, [_fp, _fp, _fp, c_int, c_int, c_int, c_void_p]),
    "ttsx_t1_create": (c_int, [POINTER(T1Weights), c_int, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "ttsx_t1_destroy": (c_int, [c_void_p]),
    "ttsx_t1_encoder_run": (c_int, [c_void_p, _fp, _fp, c_int, c_int, _fp, c_void_p]),
    "ttsx_t1_decoder_run": (c_int, [c_void_p, _fp, _fp, c_int, c_int, c_int, _fp, _fp, _fp, _fp, POINTER(c_int),
                                    c_void_p]),
    "ttsx_t1_set_plan": (c_int, [c_void_p, c_int]),
    "ttsx_t1_postnet_run": (c_int, [c_void_p, _fp, _fp, c_int, c_int, _fp, c_void_p]),
    "ttsx_spec_to_mag": (c_int, [_fp, _fp, c_int, c_int, c_int, c_int, POINTER(AudioParams), _fp, c_void_p]),
    "ttsx_griffin_lim": (c_int, [_fp, _fp, _fp, c_int, c_int, POINTER(StftParams), c_int, c_float, _fp,
                                 c_void_p]),
    "ttsx_griffin_lim_noalloc": (c_int, [_fp, _fp, _fp, c_int, c_int, POINTER(StftParams), c_int, c_float, _fp,
                                         c_void_p]),
    "ttsx_stft": (c_int, [_fp, c_int, c_int, POINTER(StftParams), _fp, _fp, c_void_p]),
    "ttsx_istft": (c_int, [_fp, _fp, c_int, c_int, POINTER(StftParams), _fp, c_void_p]),
    "ttsx_vocoder_create": (c_int, [POINTER(StftParams), c_void_p, POINTER(c_void_p)]),
    "ttsx_vocoder_destroy": (c_int, [c_void_p]),
    "ttsx_vocoder_griffin_lim": (c_int, [c_void_p, _fp, _fp, _fp, c_int, c_int, c_int, c_float, _fp, c_void_p]),
    "ttsx_vocoder_stft": (c_int, [c_void_p, _fp, c_int, c_int, _fp, _fp, c_void_p]),
    "ttsx_vocoder_istft": (c_int, [c_void_p, _fp, _fp, c_int, c_int, _fp, c_void_p]),
    "ttsx_request_tail": (c_int, [_fp, _fp, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _fp, _fp, _fp,
                                  c_long, POINTER(c_long), POINTER(c_float), c_void_p]),
    "ttsx_debug_fail_next_wait": (c_int, [ctypes.c_uint]),
}

_lib = None


class TtsxError(RuntimeError):
    pass


VERSION = 2   # TTSX_VERSION of include/ttsx.h (2: request_tail offsets [B + 2], zero-filled tail rows)


def load():
    """Load libttsx.so (once) and attach prototypes.  Raises if it is absent:
    the kernels are the product, there is nothing to fall back to."""
    global _lib
    if _lib is None:
        # torch first: its bundled HIP runtime must be the one already in the process when
        # libttsx.so resolves libamdhip64, so that both share devices, streams and allocations
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise TtsxError("%s not found: build it with `make -C tts_amd/csrc` (or "
                            "`python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.ttsx_version() != VERSION:
            raise TtsxError("%s has interface version %d, this package was written against %d: rebuild it"
                            % (LIB_PATH, lib.ttsx_version(), VERSION))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != OK:
        msg = load().ttsx_last_error_string().decode("utf8", "replace")
        if rc == ERR_ARG:
            raise ValueError("%s: %s" % (what, msg))
        raise TtsxError("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor handed to libttsx must be contiguous"
    return c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
