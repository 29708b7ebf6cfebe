"""The CPU oracle's optional Tacotron2 variants (windowing, forward attention with transition agent
and masking, attention without location features, Graves attention, speaker embedding,
inference_truncated) replayed against fixtures produced by the reference itself
(oracle/gen_golden.py t2v imports /root/reference/Tacotron2/TTS/models.py)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import t2_oracle
from tts_amd import synthetic

from conftest import GOLDEN

CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "t2v_*.npz")) if not p.endswith("t2v_truncated.npz"))
TOL = 2e-5
# Fixtures whose trajectory passes through steps where the softmax alignment jumps between two
# positions: there an fp64 evaluation differs from the fp32 reference by as much as the fp32
# restatement does (measured 9.5e-5 vs 8.7e-5), so the bound is rounding noise, not the oracle.
LOOSE = {"t2v_speaker_T44_max20": 2e-4}


def variant_setup(meta):
    """(state_dict, attention options of the oracle, speaker id or None) of a t2v fixture."""
    o = dict(meta["options"])
    speaker_id = o.pop("speaker_id", None)
    num_speakers = o.pop("num_speakers", 0)
    sd = synthetic.t2_state_dict(seed=meta["weight_seed"], gains=meta["gains"],
                                 attn_type=o.get("attn_type", "original"),
                                 location_attn=o.get("location_attn", True),
                                 trans_agent=o.get("trans_agent", False), attn_K=o.get("attn_K", 5),
                                 num_speakers=num_speakers)
    attn = dict(attn_type=o.get("attn_type", "original"), location_attn=o.get("location_attn", True),
                windowing=o.get("attn_win", False), forward_attn=o.get("forward_attn", False),
                trans_agent=o.get("trans_agent", False), forward_attn_mask=o.get("forward_attn_mask", False),
                attn_K=o.get("attn_K", 5))
    return sd, attn, speaker_id


def test_variant_fixtures_present():
    assert len(CASES) >= 9


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_variant_matches_reference_fixture(path):
    g = np.load(path)
    meta = json.loads(str(g["meta"]))
    sd, attn, speaker_id = variant_setup(meta)
    out = t2_oracle.inference(sd, g["ids"], r=1, r_init=7, attn_norm=meta["attn_norm"], prenet_type="bn",
                              max_steps=meta["max_decoder_steps"], attn=attn,
                              speaker_ids=None if speaker_id is None else [speaker_id])
    assert out["alignments"].shape == g["alignments"].shape      # same stop step
    tol = LOOSE.get(os.path.basename(path)[:-4], TOL)
    for k in ("decoder_output", "postnet_output", "alignments", "stop_tokens"):
        assert np.abs(out[k].numpy() - g[k]).max() < tol, k
    ref_al = g["alignments"][0]
    srt = np.sort(ref_al, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-4
    assert (out["alignments"][0].numpy().argmax(-1)[clear] == ref_al.argmax(-1)[clear]).all()


def test_oracle_truncated_matches_reference_fixture():
    g = np.load(os.path.join(GOLDEN, "t2v_truncated.npz"))
    meta = json.loads(str(g["meta"]))
    sd = synthetic.t2_state_dict(seed=meta["weight_seed"], gains=meta["gains"])
    orc = t2_oracle.TruncatedOracle(sd, attn_norm=meta["attn_norm"], max_steps=meta["max_decoder_steps"])
    for i in range(meta["n_calls"]):
        out = orc(g["ids%d" % i])
        for k in ("decoder_output", "postnet_output", "alignments", "stop_tokens"):
            want = g["%s%d" % (k, i)]
            assert out[k].shape == want.shape, (i, k)
            assert np.abs(out[k].numpy() - want).max() < TOL, (i, k)
