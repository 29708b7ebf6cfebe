"""The C-ABI library loads and exports every symbol include/ttsx.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from tts_amd import _lib
    header = open(os.path.join(ROOT, "include", "ttsx.h")).read()
    declared = set(re.findall(r"\b(ttsx_[a-z0-9_]+)\s*\(", header))
    declared -= {"ttsx_t2_decoder_weights", "ttsx_postnet_weights", "ttsx_encoder_conv_weights"}
    assert len(declared) >= 15
    assert os.path.exists(_lib.LIB_PATH), "build libttsx.so first (make -C tts_amd/csrc)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.ttsx_version() == 2


def test_struct_layouts_match_header():
    from tts_amd import _lib
    # 5 pairs of prenet pointers + 18 single pointers
    assert ctypes.sizeof(_lib.T2DecoderWeights) == 8 * (10 + 18)
    assert ctypes.sizeof(_lib.PostnetWeights) == 8 * 30
    assert ctypes.sizeof(_lib.EncoderConvWeights) == 8 * 18
    assert ctypes.sizeof(_lib.CbhgWeights) == 16 + 8 * 115
    assert ctypes.sizeof(_lib.T1Weights) == 2 * (16 + 8 * 115) + 8 * 39
    assert ctypes.sizeof(_lib.EncoderLstmWeights) == 8 * 8
    assert ctypes.sizeof(_lib.T2AttentionOptions) == 16 + 16 + 8 + 32   # 4 ints, 2 pointers, int + pad, 4 pointers
    assert ctypes.sizeof(_lib.StftParams) == 16
    assert ctypes.sizeof(_lib.AudioParams) == 28


def test_no_cpu_fallback_in_product_path():
    """The package must not import the oracle, and must fail loudly without a GPU."""
    import torch
    import pytest
    pkg = os.path.join(ROOT, "tts_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn
    if not torch.cuda.is_available():
        from tts_amd import _lib, synthetic
        from tts_amd.tacotron2 import Tacotron2
        m = Tacotron2(num_chars=129, num_speakers=0, r=7, attn_norm="sigmoid", prenet_type="bn")
        m.load_state_dict(synthetic.t2_state_dict(0))
        with pytest.raises(_lib.TtsxError):
            m.inference(torch.zeros(1, 5, dtype=torch.long))


def test_entry_points_reject_bad_arguments_before_any_device_call():
    """Null handles / pointers and out-of-range sizes come back as TTSX_ERR_ARG with a message naming the entry
    point; the check precedes every HIP call, so it runs (and must not crash) on a machine without a GPU."""
    from ctypes import byref, c_int, c_long, c_float, c_void_p
    from tts_amd import _lib
    lib = _lib.load()
    one = c_void_p(256)   # a non-null pointer that is never dereferenced on these paths
    out = c_void_p()
    n = c_int()
    sp = _lib.StftParams(2048, 275, 1100, 0)
    calls = {
        "ttsx_t2_decoder_create": lambda: lib.ttsx_t2_decoder_create(None, 7, 0, None, byref(out)),
        "ttsx_t2_decoder_set_attention": lambda: lib.ttsx_t2_decoder_set_attention(None, None, None),
        "ttsx_t2_decoder_set_initial": lambda: lib.ttsx_t2_decoder_set_initial(None, 0, None, 1, None),
        "ttsx_t2_decoder_run": lambda: lib.ttsx_t2_decoder_run(None, one, None, 1, 8, 1, 10, 0.5, 0, one, one, one,
                                                               one, byref(n), None),
        "ttsx_t2_decoder_set_plan": lambda: lib.ttsx_t2_decoder_set_plan(None, -1, -1, -1),
        "ttsx_t2_decoder_set_team": lambda: lib.ttsx_t2_decoder_set_team(None, -1),
        "ttsx_t2_decoder_set_fusion": lambda: lib.ttsx_t2_decoder_set_fusion(None, -1),
        "ttsx_t2_decoder_set_pieces": lambda: lib.ttsx_t2_decoder_set_pieces(None, -1),
        "ttsx_t2_decoder_set_polling": lambda: lib.ttsx_t2_decoder_set_polling(None, 0, 0),
        "ttsx_t2_decoder_steps_launched": lambda: lib.ttsx_t2_decoder_steps_launched(None),
        "ttsx_t2_decoder_begin": lambda: lib.ttsx_t2_decoder_begin(None, one, None, 1, 8, None),
        "ttsx_t2_decoder_load_state": lambda: lib.ttsx_t2_decoder_load_state(None, *([None] * 7), None),
        "ttsx_t2_decoder_step": lambda: lib.ttsx_t2_decoder_step(None, one, 1, one, one, one, None),
        "ttsx_t2_decoder_read_state": lambda: lib.ttsx_t2_decoder_read_state(None, *([None] * 7), None),
        "ttsx_postnet_create": lambda: lib.ttsx_postnet_create(None, None, byref(out)),
        "ttsx_postnet_run": lambda: lib.ttsx_postnet_run(None, one, None, 1, 8, one, None),
        "ttsx_encoder_convs_create": lambda: lib.ttsx_encoder_convs_create(None, None, byref(out)),
        "ttsx_encoder_convs_run": lambda: lib.ttsx_encoder_convs_run(None, one, None, 1, 8, one, None),
        "ttsx_encoder_lstm_create": lambda: lib.ttsx_encoder_lstm_create(None, None, byref(out)),
        "ttsx_encoder_lstm_run": lambda: lib.ttsx_encoder_lstm_run(None, one, None, 1, 8, one, None),
        "ttsx_encoder_lstm_run_stateful": lambda: lib.ttsx_encoder_lstm_run_stateful(None, one, None, 1, 8, one, None,
                                                                                     None, None),
        "ttsx_encoder_lstm_set_plan": lambda: lib.ttsx_encoder_lstm_set_plan(None, -1),
        "ttsx_add_speaker_embedding": lambda: lib.ttsx_add_speaker_embedding(one, one, None, 4, 1, 8, None),
        "ttsx_t1_create": lambda: lib.ttsx_t1_create(None, 5, 5, 1025, None, byref(out)),
        "ttsx_t1_encoder_run": lambda: lib.ttsx_t1_encoder_run(None, one, None, 1, 8, one, None),
        "ttsx_t1_decoder_run": lambda: lib.ttsx_t1_decoder_run(None, one, None, 1, 8, 10, one, one, one, one, byref(n),
                                                               None),
        "ttsx_t1_set_plan": lambda: lib.ttsx_t1_set_plan(None, -1),
        "ttsx_t1_postnet_run": lambda: lib.ttsx_t1_postnet_run(None, one, None, 1, 8, one, None),
        "ttsx_spec_to_mag": lambda: lib.ttsx_spec_to_mag(None, None, 1, 8, 80, 1025, None, one, None),
        "ttsx_griffin_lim": lambda: lib.ttsx_griffin_lim(None, one, None, 1, 8, byref(sp), 60, 1.0, one, None),
        "ttsx_griffin_lim_noalloc": lambda: lib.ttsx_griffin_lim_noalloc(None, one, None, 1, 8, byref(sp), 60, 1.0, one,
                                                                         None),
        "ttsx_stft": lambda: lib.ttsx_stft(None, 1, 8, byref(sp), one, one, None),
        "ttsx_istft": lambda: lib.ttsx_istft(None, one, 1, 8, byref(sp), one, None),
        "ttsx_vocoder_create": lambda: lib.ttsx_vocoder_create(byref(sp), None, None),
        "ttsx_vocoder_griffin_lim": lambda: lib.ttsx_vocoder_griffin_lim(None, one, one, None, 1, 8, 60, 0.98, one, None),
        "ttsx_vocoder_stft": lambda: lib.ttsx_vocoder_stft(None, one, 1, 8, one, one, None),
        "ttsx_vocoder_istft": lambda: lib.ttsx_vocoder_istft(None, one, one, 1, 8, one, None),
        "ttsx_request_tail": lambda: lib.ttsx_request_tail(None, None, 1, 100, 1024, 256, -40.0, 0, 0, one, one, one,
                                                           c_long(100), None, None, None),
        "ttsx_debug_fail_next_wait": lambda: lib.ttsx_debug_fail_next_wait(16),
    }
    skipped = {"ttsx_last_error_string", "ttsx_version"}
    destroys = {k for k in _lib.PROTOTYPES if k.endswith("_destroy")}
    assert set(calls) | skipped | destroys == set(_lib.PROTOTYPES)
    for name, call in calls.items():
        assert call() == _lib.ERR_ARG, name
        msg = lib.ttsx_last_error_string().decode()
        assert msg, name
    for name in destroys:       # destroying nothing is not an error (free(NULL) semantics)
        assert getattr(lib, name)(None) == _lib.OK, name
    # shapes and ranges
    bad_sp = _lib.StftParams(1024, 275, 1100, 0)
    assert lib.ttsx_stft(one, 1, 8, byref(bad_sp), one, one, None) == _lib.ERR_ARG
    assert "n_fft" in lib.ttsx_last_error_string().decode()
    assert lib.ttsx_stft(one, 1, 1, byref(sp), one, one, None) == _lib.ERR_ARG      # a single frame
    assert lib.ttsx_griffin_lim(one, one, None, 1, 8, byref(sp), -1, 1.0, one, None) == _lib.ERR_ARG
    assert lib.ttsx_griffin_lim_noalloc(one, one, None, 64, 800, byref(sp), 60, 0.98, one, None) == _lib.ERR_ARG
    assert "static scratch" in lib.ttsx_last_error_string().decode()               # beyond the fixed maximum
    assert lib.ttsx_postnet_run(one, one, None, 0, 8, one, None) == _lib.ERR_ARG    # empty batch
    assert lib.ttsx_encoder_convs_run(one, one, None, 1, 0, one, None) == _lib.ERR_ARG   # empty text


def test_header_is_plain_c_and_links_from_a_c_host(tmp_path):
    """include/ttsx.h compiles as C99 (-pedantic) and a C program linked against libttsx.so can call it: the
    boundary is a C ABI, not a C++ or Python one."""
    import shutil
    import subprocess
    from tts_amd import _lib
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "host.c"
    src.write_text('#include <stdio.h>\n#include "ttsx.h"\n'
                   "int main(void) {\n"
                   "  int rc = ttsx_postnet_run(NULL, NULL, NULL, 1, 1, NULL, NULL);\n"
                   '  printf("%d %d %s\\n", ttsx_version(), rc, ttsx_last_error_string());\n'
                   "  return rc == TTSX_ERR_ARG ? 0 : 1;\n}\n")
    exe = tmp_path / "host"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                         str(src), "-L", libdir, "-lttsx", "-Wl,-rpath," + libdir, "-o", str(exe)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ttsx_postnet_run" in run.stdout
