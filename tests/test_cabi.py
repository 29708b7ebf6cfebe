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
    assert lib.ttsx_version() >= 1


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
