"""Config-file fidelity: tts_amd/ref_configs.json holds what the REFERENCE's own loaders return for its three
config files (oracle/gen_golden.py config); the package's parser, model factory and AudioProcessor must give
the same configuration from a file in the same format."""
import json
import os

import pytest

from conftest import ROOT

REF = json.load(open(os.path.join(ROOT, "tts_amd", "ref_configs.json")))
MODEL_CONFIGS = ("Tacotron2/models/config.json", "Tacotron/models/config.json")


def _dumps(v):
    # a literal `//` inside a string would start a comment for this dialect: the reference's files escape the
    # slashes (tcp:\/\/localhost), which JSON reads back as plain slashes
    return json.dumps(v).replace("//", "\\/\\/")


def _as_commented_file(parsed):
    """A file in the reference's config dialect for `parsed`: `//` comments after values and on lines of their
    own, and a backslash-newline continuation inside a long string."""
    lines = ["{   // model configuration"]
    items = list(parsed.items())
    for i, (k, v) in enumerate(items):
        comma = "," if i + 1 < len(items) else ""
        if isinstance(v, dict):
            lines.append('    "%s": {' % k)
            sub = list(v.items())
            for j, (sk, sv) in enumerate(sub):
                lines.append('        "%s": %s%s   // %s of %s' % (sk, _dumps(sv), "," if j + 1 < len(sub) else "", sk, k))
            lines.append("    }%s" % comma)
        elif isinstance(v, str) and len(v) > 12 and "//" not in v:
            half = len(v) // 2
            lines.append('    "%s": %s\\\n%s%s  // continued line' % (k, _dumps(v[:half])[:-1], _dumps(v[half:])[1:], comma))
        else:
            lines.append('    "%s": %s%s // %s' % (k, _dumps(v), comma, k))
        if i % 7 == 3:
            lines.append("    // a comment on a line of its own")
    lines.append("}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("name", MODEL_CONFIGS)
def test_load_config_reproduces_what_the_reference_parsed(tmp_path, name):
    from tts_amd.config import load_config
    parsed = REF[name]["parsed"]
    path = tmp_path / "config.json"
    text = _as_commented_file(parsed)
    assert "//" in text and "\\\n" in text and "tcp:\\/\\/" in text + ("tcp:\\/\\/" if "distributed" not in parsed else "")
    path.write_text(text)
    cfg = load_config(str(path))
    assert dict(cfg) == parsed
    assert cfg.r == parsed["r"] and cfg.audio["num_mels"] == 80        # attribute access like the reference's AttrDict
    with pytest.raises(AttributeError):
        cfg.no_such_key


def test_server_config_map_semantics(tmp_path):
    from tts_amd.config import load_server_config, repo_config
    parsed = REF["Tacotron2/conf.json"]["parsed"]
    path = tmp_path / "conf.json"
    path.write_text(json.dumps(parsed))
    args = load_server_config(str(path))
    assert dict(args) == parsed == dict(repo_config("Tacotron2/conf.json"))
    assert args.tts_config == "models/config.json" and args.use_cuda is False and args.port == 5002
    assert REF["Tacotron2/conf.json"]["missing_key_is_none"] and args.wavernn_path is None
    args.use_cuda = True
    assert args["use_cuda"] is True


def test_repo_configs_build_the_right_models_and_audio_processors(capsys):
    """setup_model(129, 0, cfg) + AudioProcessor(**cfg.audio) as Synthesizer.load_tts does it
    (Tacotron2/TTS/synthesizer.py:48-62): architecture switches and STFT geometry."""
    from tts_amd.audio import AudioProcessor
    from tts_amd.config import repo_config
    from tts_amd.tacotron2 import setup_model
    c2 = repo_config("Tacotron2/models/config.json")
    m = setup_model(129, 0, c2)
    assert m.decoder.r == m.decoder.r_init == c2.r == 7
    assert (m.attn_norm, m.prenet_type, m.attn_type) == ("sigmoid", "bn", "original")
    assert m.location_attn and not (m.attn_win or m.forward_attn or m.trans_agent or m.forward_attn_mask)
    assert m.decoder.max_decoder_steps == 1000
    for name, sym, max_norm in (("Tacotron2/models/config.json", True, 4), ("Tacotron/models/config.json", False, 1)):
        cfg = repo_config(name)
        v1 = name.startswith("Tacotron/")
        ap = AudioProcessor(**cfg.audio, **({"tacotron_v1": True} if v1 else {}))
        d = REF[name]["derived"]
        assert (ap.n_fft, ap.hop_length, ap.win_length) == (d["n_fft"], d["hop_length"], d["win_length"])
        assert d["win_length"] == (1102 if v1 else 1100)
        assert ap.symmetric_norm is sym and ap.max_norm == float(max_norm) and ap.griffin_lim_iters == 60
        assert ap.power == 1.5 and ap.preemphasis == 0.98 and ap.sample_rate == 22050
    c1 = repo_config("Tacotron/models/config.json")
    assert (c1.r, c1.memory_size, c1.embedding_size) == (2, 5, 256)
