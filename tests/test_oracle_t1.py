"""The Tacotron (v1) CPU oracle replayed against fixtures produced by the reference itself
(oracle/gen_golden.py t1 imports /root/reference/Tacotron/TTS/models.py)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import t1_oracle
from tts_amd import synthetic

from conftest import GOLDEN

CASES = sorted(glob.glob(os.path.join(GOLDEN, "t1_T*.npz")))


def test_fixtures_present():
    assert len(CASES) >= 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_fixture(path):
    g = np.load(path)
    meta = json.loads(str(g["meta"]))
    sd = synthetic.t1_state_dict(seed=meta["weight_seed"])
    out = t1_oracle.forward(sd, g["ids"], r=meta["r"], memory_size=meta["memory_size"],
                            max_decoder_steps=meta["max_decoder_steps"])
    assert out["alignments"].shape == g["alignments"].shape      # exact stop step
    assert out["mel"].shape == g["mel"].shape and out["linear"].shape == g["linear"].shape
    for k in ("mel", "linear", "alignments", "stop_tokens"):
        err = np.abs(out[k].numpy() - g[k]).max()
        assert err < 2e-5, (k, err)
