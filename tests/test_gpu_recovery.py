"""What a run does after a bounded in-launch wait has expired.

Some launches hold workgroups that wait for other workgroups of the same launch.  On an idle GPU such a wait
never expires, so these tests ask the library to report one (ttsx_debug_fail_next_wait): under a default plan the run
must start over on the plain launch chain and return that chain's results; under a plan forced by its setter it must
fail with TTSX_ERR_STATE."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STEP_FUSION, T2_TEAM, T1_TEAM, ENCODER_SCAN = 1, 2, 4, 8


def _fail_next(kinds):
    from tts_amd import _lib
    _lib.check(_lib.load().ttsx_debug_fail_next_wait(kinds), "ttsx_debug_fail_next_wait")


def _t2_model():
    from tts_amd import synthetic
    from tts_amd.tacotron2 import Tacotron2
    m = Tacotron2(num_chars=129, num_speakers=0, r=7, attn_norm="sigmoid", prenet_type="bn", prenet_dropout=False,
                  separate_stopnet=True)
    m.load_state_dict(synthetic.t2_state_dict(seed=0, prenet_type="bn", gains={"stop_bias": -30.0}))
    m.cuda().eval()
    m.decoder.set_r(1)
    return m


def _decode(m, enc, lens, steps):
    mel, align, stop, _, run = m.decode(enc, lens, fixed_steps=steps)
    assert run == steps
    return [t.cpu().numpy() for t in (mel, align, stop)]


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_batched_step_without_its_fusions_after_an_expired_wait():
    from tts_amd import _lib, synthetic
    B, T, steps = 64, 100, 40
    m = _t2_model()
    lens = torch.as_tensor([max(1, T - 3 * (b % 7)) for b in range(B)])
    enc = m.encode(torch.as_tensor(synthetic.t2_text_ids(3, B, T)).cuda(), lens)
    m.set_fusion(0)
    plain = _decode(m, enc, lens, steps)
    m.set_fusion(-1)
    _fail_next(STEP_FUSION)
    assert _same(_decode(m, enc, lens, steps), plain)     # started over, fusions off
    assert _same(_decode(m, enc, lens, steps), plain)     # and the handle stays that way
    m.set_fusion(2)
    assert _same(_decode(m, enc, lens, steps), plain)     # (forced on: the fused launches give the same bits)
    _fail_next(STEP_FUSION)
    with pytest.raises(_lib.TtsxError, match="waited too long"):
        m.decode(enc, lens, fixed_steps=steps)
    assert _same(_decode(m, enc, lens, steps), plain)     # the failure was reported once


def test_one_utterance_on_the_launch_chain_after_the_team_gave_up():
    from tts_amd import _lib, synthetic
    T, steps = 60, 50
    m = _t2_model()
    enc = m.encode(torch.as_tensor(synthetic.t2_text_ids(4, 1, T)).cuda())
    m.set_team(0)
    chain = _decode(m, enc, None, steps)
    m.set_team(1)
    team = _decode(m, enc, None, steps)
    assert not _same(team, chain)                          # (sums grouped differently: equal to rounding only)
    assert max(np.abs(x - y).max() for x, y in zip(team, chain)) < 1e-4
    m.set_team(-1)
    assert _same(_decode(m, enc, None, steps), team)       # the default at this shape is the team
    _fail_next(T2_TEAM)
    assert _same(_decode(m, enc, None, steps), chain)      # started over on the chain
    assert _same(_decode(m, enc, None, steps), chain)      # cooling down: still the chain
    m.set_team(1)
    _fail_next(T2_TEAM)
    with pytest.raises(_lib.TtsxError, match="waited too long"):
        m.decode(enc, None, fixed_steps=steps)
    assert _same(_decode(m, enc, None, steps), team)


def test_tacotron_v1_on_the_launch_chain_after_the_team_gave_up():
    from tts_amd import _lib, synthetic
    from tts_amd.tacotron import Tacotron
    m = Tacotron(61, 256, 1025, 80, 2, None, 5)
    m.load_state_dict(synthetic.t1_state_dict(seed=0, r=2, memory_size=5))
    m.cuda().eval()
    m.decoder.max_decoder_steps = 40
    lens = [57]
    enc = m.encode(torch.as_tensor(synthetic.t1_text_ids(61, 57)).cuda(), lens)

    def run():
        return [t.cpu().numpy() for t in m.decode(enc, lens)]

    m.set_plan(0)
    chain = run()
    m.set_plan(1)
    team = run()
    assert not _same(team, chain) and (team[3] == chain[3]).all()
    m.set_plan(-1)
    assert _same(run(), team)
    _fail_next(T1_TEAM)
    assert _same(run(), chain)
    assert _same(run(), chain)                             # cooling down
    m.set_plan(1)
    _fail_next(T1_TEAM)
    with pytest.raises(_lib.TtsxError, match="waited too long"):
        m.decode(enc, lens)
    assert _same(run(), team)


def test_encoder_recurrence_as_step_launches_after_the_scan_gave_up():
    from tts_amd import _lib, synthetic
    m = _t2_model()
    ids = torch.as_tensor(synthetic.t2_text_ids(6, 1, 90)).cuda()
    m.set_encoder_plan(0)
    chain = m.encode(ids).cpu().numpy()
    m.set_encoder_plan(1)
    scan = m.encode(ids).cpu().numpy()
    assert not np.array_equal(scan, chain) and np.abs(scan - chain).max() < 2e-6   # (equal to rounding only)
    m.set_encoder_plan(-1)
    assert np.array_equal(m.encode(ids).cpu().numpy(), scan)    # the default at batch 1 is the scan
    _fail_next(ENCODER_SCAN)
    assert np.array_equal(m.encode(ids).cpu().numpy(), chain)   # started over as step launches
    assert np.array_equal(m.encode(ids).cpu().numpy(), chain)   # cooling down
    m.set_encoder_plan(1)
    _fail_next(ENCODER_SCAN)
    with pytest.raises(_lib.TtsxError, match="waited too long"):
        m.encode(ids)
    assert np.array_equal(m.encode(ids).cpu().numpy(), scan)
    m.set_encoder_plan(-1)
