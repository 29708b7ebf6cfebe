"""Config helpers with the reference's behaviour (Tacotron2/TTS/utils.py:1007-1048).

`ref_configs.json` next to this file holds the PARSED content of the reference's three config files
(Tacotron2/models/config.json, Tacotron/models/config.json, Tacotron2/conf.json) as the reference's own
loaders return them (written by the golden-vector generator, which calls those loaders in the build container).
"""
import json
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))


class AttrDict(dict):
    """dict whose keys are attributes too (utils.py:1007-1010): a missing key raises."""

    def __init__(self, *args, **kwargs):
        super(AttrDict, self).__init__(*args, **kwargs)
        self.__dict__ = self


class Map(dict):
    """The server's conf.json wrapper (utils.py:1012-1038): attribute access, a missing key reads as None
    (Synthesizer relies on that for `wavernn_path`, Tacotron2/TTS/synthesizer.py:37-39)."""

    def __init__(self, *args, **kwargs):
        super(Map, self).__init__()
        for src in list(args) + [kwargs]:
            if isinstance(src, dict):
                for k, v in src.items():
                    self[k] = v

    def __getattr__(self, attr):
        return self.get(attr)

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]


_CONTINUATION = re.compile(r"\\\n")
_COMMENT = re.compile(r"//.*\n")


def parse_config(text):
    """JSON with `//` comments and backslash-newline continuations (utils.py:1040-1048).  As in the reference
    a comment runs from `//` to the end of its line wherever the `//` stands - also inside a string value,
    which is why the reference's config writes its URL with escaped slashes (tcp:\\/\\/localhost)."""
    return AttrDict(json.loads(_COMMENT.sub("\n", _CONTINUATION.sub("", text))))


def load_config(config_path):
    with open(config_path, "r") as f:
        return parse_config(f.read())


def load_server_config(path):
    """conf.json as Tacotron2/Tacotron2.py:41-42 reads it: plain JSON into a Map."""
    with open(path, "rt") as f:
        return Map(json.load(f))


def _ref_configs():
    with open(os.path.join(_HERE, "ref_configs.json")) as f:
        return json.load(f)


def repo_config(name="Tacotron2/models/config.json"):
    """The parsed content of one of the reference's config files: 'Tacotron2/models/config.json' and
    'Tacotron/models/config.json' as AttrDict, 'Tacotron2/conf.json' as Map."""
    parsed = _ref_configs()[name]["parsed"]
    return Map(parsed) if name.endswith("conf.json") else AttrDict(parsed)


def tacotron2_repo_config():
    """The repo's Tacotron2 model config (Tacotron2/models/config.json) as `load_config` returns it."""
    return repo_config("Tacotron2/models/config.json")
