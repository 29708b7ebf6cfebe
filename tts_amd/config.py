"""Config helpers with the reference's behaviour (Tacotron2/TTS/utils.py:1007-1048)."""
import json
import re


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super(AttrDict, self).__init__(*args, **kwargs)
        self.__dict__ = self


def load_config(config_path):
    """JSON with `//` comments and backslash-newline continuations, as the reference parses it
    (utils.py:1040-1048)."""
    config = AttrDict()
    with open(config_path, "r") as f:
        input_str = f.read()
    input_str = re.sub(r'\\\n', '', input_str)
    input_str = re.sub(r'//.*\n', '\n', input_str)
    config.update(json.loads(input_str))
    return config


def tacotron2_repo_config():
    """The hyper-parameters of the repo's Tacotron2 model config that the serving path reads
    (values of Tacotron2/models/config.json:9-29, 39, 81-92)."""
    return AttrDict(
        model="Tacotron2", r=7, prenet_type="bn", prenet_dropout=False, attention_type="original",
        attention_heads=5, attention_norm="sigmoid", windowing=False, use_forward_attn=False,
        forward_attn_mask=False, transition_agent=False, location_attn=True, bidirectional_decoder=False,
        separate_stopnet=True, use_gst=False, use_phonemes=True, enable_eos_bos_chars=False,
        audio=dict(num_mels=80, num_freq=1025, sample_rate=22050, frame_length_ms=50, frame_shift_ms=12.5,
                   preemphasis=0.98, min_level_db=-100, ref_level_db=20, power=1.5, griffin_lim_iters=60,
                   signal_norm=True, symmetric_norm=True, max_norm=4, clip_norm=True, mel_fmin=0.0,
                   mel_fmax=8000.0, do_trim_silence=True))
