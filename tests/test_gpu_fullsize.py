"""The benchmark configuration at full size (64 sentences x 128 ids, 800 decoder steps, r = 1,
51 200 frames, 60 Griffin-Lim iterations), checked through properties that do not need the CPU
oracle to run at that size: rows that hold the same utterance are bit-equal wherever they sit in
the batch, runs are bit-reproducible, Griffin-Lim is exactly equivariant to a power-of-two scale
of the magnitudes, STFT -> ISTFT reproduces the signal; plus a bounded comparison of the first
steps / one utterance against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T_IN, STEPS = 64, 128, 800


@pytest.fixture(scope="module")
def model():
    import bench
    m, ap, sd = bench.build_model(0, 1)
    return m, ap, sd


@pytest.fixture(scope="module")
def decoded(model):
    from tts_amd import synthetic
    m, ap, sd = model
    base = synthetic.t2_text_ids(4242, 8, T_IN)
    ids = np.concatenate([base] * 8)                      # utterance b = base[b % 8]
    out = m.inference_batch(torch.as_tensor(ids), fixed_steps=STEPS)
    return ids, {k: v for k, v in out.items()}


def test_decoder_rows_with_the_same_text_are_bit_equal(decoded):
    ids, out = decoded
    mel, align, post = out["decoder_output"], out["alignments"], out["postnet_output"]
    assert mel.shape == (B, STEPS, 80) and align.shape == (B, STEPS, T_IN) and post.shape == (B, STEPS, 80)
    assert torch.isfinite(mel).all() and torch.isfinite(post).all()
    for b in range(8, B):
        assert torch.equal(mel[b], mel[b % 8]) and torch.equal(align[b], align[b % 8])
        assert torch.equal(post[b], post[b % 8])
    # rows of different texts differ, alignments are distributions
    assert not torch.equal(mel[0], mel[1])
    assert (align.sum(-1) - 1).abs().max() < 1e-5 and align.min() >= 0


def test_decoder_is_reproducible_and_agrees_with_other_tilings_and_the_oracle(model, decoded):
    from oracle import t2_oracle
    m, ap, sd = model
    ids, out = decoded
    again = m.inference_batch(torch.as_tensor(ids), fixed_steps=STEPS)
    for k in ("decoder_output", "alignments", "postnet_output"):
        assert torch.equal(again[k], out[k]), k
    # the same texts as a batch of 16 (another MFMA tiling) and as batch 1 (the GEMV engine), 40 steps
    n = 40
    small = m.inference_batch(torch.as_tensor(ids[:16]), fixed_steps=n)
    assert (small["decoder_output"] - out["decoder_output"][:16, :n]).abs().max() < 5e-4
    one = m.inference_batch(torch.as_tensor(ids[3:4]), fixed_steps=n)
    assert (one["decoder_output"][0] - out["decoder_output"][3, :n]).abs().max() < 5e-4
    orc = t2_oracle.inference(sd, ids[3:4], r=1, attn_norm="sigmoid", prenet_type="bn", max_steps=n, fixed_steps=True)
    assert (out["decoder_output"][3, :n].cpu() - orc["decoder_output"][0]).abs().max() < 1e-3
    assert (out["alignments"][3, :n].cpu() - orc["alignments"][0]).abs().max() < 1e-3


def test_postnet_full_size_against_oracle_on_one_utterance(model, decoded):
    from oracle import t2_oracle
    m, ap, sd = model
    ids, out = decoded
    # all 51 200 frames unmasked (inference_batch masks each row at its own stop index)
    post = m.postnet(out["decoder_output"])
    for b in range(8, B):
        assert torch.equal(post[b], post[b % 8])
    mel = out["decoder_output"][5:6].cpu()
    ref = t2_oracle.postnet(sd, mel.transpose(1, 2)).transpose(1, 2)
    assert (post[5].cpu() - ref[0]).abs().max() < 1e-3
    # the masked output agrees with it up to the row's stop index (minus the conv reach at the cut)
    n = int(out["n_steps"][5])
    assert torch.equal(out["postnet_output"][5, :max(n - 10, 0)], post[5, :max(n - 10, 0)])


def test_griffin_lim_full_size_scale_equivariance_batch_invariance_reproducibility(model, decoded):
    m, ap, sd = model
    ids, out = decoded
    mag = ap.spec_to_mag(out["postnet_output"])
    gen = torch.Generator(device="cuda").manual_seed(11)
    phase = torch.rand(B, STEPS, 1025, device="cuda", generator=gen)
    phase[8:] = phase[:8].repeat(7, 1, 1)
    y = ap.griffin_lim_batch(mag, phase)
    assert y.shape == (B, 275 * (STEPS - 1)) and torch.isfinite(y).all() and float(y.abs().max()) > 0
    assert torch.equal(ap.griffin_lim_batch(mag, phase), y)                       # reproducible
    for b in range(8, B):
        assert torch.equal(y[b], y[b % 8])                                        # position in the batch is irrelevant
    assert torch.equal(ap.griffin_lim_batch(mag[2:3].contiguous(), phase[2:3].contiguous())[0], y[2])
    # every operation is linear in the magnitudes or invariant to their scale: a factor 2 is exact
    assert torch.equal(ap.griffin_lim_batch(2.0 * mag, phase), 2.0 * y)


def test_stft_istft_round_trip_at_full_length(model):
    m, ap, sd = model
    rng = np.random.default_rng(0)
    y = (rng.standard_normal(275 * (STEPS - 1)) * 0.1).astype(np.float32)
    X = ap._stft(y)
    assert X.shape == (1025, STEPS)
    y2 = ap._istft(X)
    assert y2.shape == y.shape
    # interior samples (covered by four full windows) come back to fp32 accuracy
    assert np.abs(y2[2048:-2048] - y[2048:-2048]).max() < 1e-4
