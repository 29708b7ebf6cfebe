"""The pure-Python text front-end (tts_amd/text.py) against cases produced by the reference's own
functions (oracle/gen_golden.py text imports /root/reference/Tacotron2/TTS/utils.py).  Grapheme to
phoneme conversion itself (phonemizer + espeak) is absent from the build container and the GPU box:
its parity is unpinned; what is pinned is everything around that call."""
import json
import os

import pytest

from tts_amd import text as T

from conftest import GOLDEN

G = json.load(open(os.path.join(GOLDEN, "text_cases.json"), encoding="utf8"))


def test_symbol_tables():
    assert T.symbols == G["symbols"] and T.phonemes == G["phonemes"]
    assert len(T.phonemes) == 129          # num_chars of the Tacotron2 model (Tacotron2/TTS/synthesizer.py:51)


@pytest.mark.parametrize("case", G["lines"], ids=[str(i) for i in range(len(G["lines"]))])
def test_cleaners_numbers_and_ids(case):
    line = case["text"]
    assert T.basic_cleaners(line) == case["basic"]
    if line.isascii():
        assert T.english_cleaners(line) == case["english_ascii"]
    else:   # transliteration needs unidecode (absent): the rest of the pipeline, as the fixture was made
        assert T.collapse_whitespace(T.expand_abbreviations(T.expand_numbers(T.lowercase(line)))) == case["english_ascii"]
    assert T.normalize_numbers(line) == case["numbers"]
    assert T.expand_abbreviations(line) == case["abbreviations"]
    ids = T.text_to_sequence(line, ["basic_cleaners"])
    assert ids == case["basic_ids"]
    assert T.sequence_to_text(ids) == case["basic_text"]


@pytest.mark.parametrize("case", G["phoneme_lines"], ids=[str(i) for i in range(len(G["phoneme_lines"]))])
def test_phoneme_strings(case):
    ids = T.phoneme_string_to_sequence(case["ph"])
    assert ids == case["ids"]
    assert T.phoneme_string_to_sequence(case["ph"], enable_eos_bos=True) == case["padded"]
    assert T.sequence_to_phoneme(ids) == case["text"]


@pytest.mark.parametrize("case", G["g2p"], ids=[str(i) for i in range(len(G["g2p"]))])
def test_punctuation_restored_after_g2p(case):
    assert T.text2phone(case["text"], "en-us", g2p=lambda t, lang: case["raw"]) == case["ph"]
    ids = T.phoneme_to_sequence(case["text"], ["basic_cleaners"], "en-us", True, g2p=lambda t, lang: case["raw"])
    assert ids[0] == T.phonemes.index("^") and ids[-1] == T.phonemes.index("~")


def test_unknown_cleaner_and_missing_backends():
    with pytest.raises(Exception):
        T.text_to_sequence("x", ["no_such_cleaner"])
    with pytest.raises(RuntimeError):
        T.convert_to_ascii("naïve")          # unidecode is not installed here
    with pytest.raises(RuntimeError):
        T.phoneme_to_sequence("hello", ["basic_cleaners"], "en-us")   # phonemizer is not installed here
