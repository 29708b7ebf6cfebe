"""Text -> id sequences for the Tacotron2 models: the pure-Python part of the reference's text
front-end (Tacotron2/TTS/utils.py:410-743).

What is here: the two symbol inventories (characters + "@"-prefixed phoneme symbols, 129 phoneme
ids), `text_to_sequence` with the curly-brace escape, `sequence_to_text`, `sequence_to_phoneme`,
the cleaners, the number speller, and the phoneme-string half of `phoneme_to_sequence`.

What is not: grapheme-to-phoneme conversion.  The reference shells out to espeak through the
`phonemizer` package (utils.py:446-468); `phoneme_to_sequence` here takes that step as a callable
(`g2p=`), uses `phonemizer` when it is importable, and raises otherwise.  `convert_to_ascii`
likewise uses `unidecode` when it is importable and otherwise only lets ASCII text through.
"""
import re

PAD, EOS, BOS = "_", "~", "^"
_LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
_PUNCT = "!'(),-.:;? "
_PHONEME_PUNCT = ".!;:,?"
_IPA_GROUPS = (
    "iyɨʉɯuɪʏʊeøɘəɵɤoɛœɜɞʌɔæɐaɶɑɒᵻ",                                   # vowels
    "ʘɓǀɗǃʄǂɠǁʛ",                                                          # non-pulmonic consonants
    "pbtdʈɖcɟkɡqɢʔɴŋɲɳnɱmʙrʀⱱɾɽɸβfvθðszʃʒʂʐçʝxɣχʁħʕhɦɬɮʋɹɻjɰlɭʎʟ",  # pulmonic consonants
    "ˈˌːˑ",                                                                # suprasegmentals
    "ʍwɥʜʢʡɕʑɺɧ",                                                          # other symbols
    "ɚ˞ɫ",                                                                 # diacritics
)
_IPA = sorted("".join(_IPA_GROUPS))

# id tables (utils.py:431-432): ids 0..2 are pad / eos / bos in both
symbols = [PAD, EOS, BOS] + list(_LETTERS + _PUNCT) + ["@" + p for p in _IPA]
phonemes = [PAD, EOS, BOS] + _IPA + list(_PUNCT)


class _Inventory:
    """A symbol <-> id table; the three control symbols never come out of text."""

    def __init__(self, table):
        self.table = list(table)
        self.index = {s: i for i, s in enumerate(self.table)}

    def encode(self, items):
        return [self.index[s] for s in items if s in self.index and s not in (PAD, EOS, BOS)]

    def decode(self, ids):
        return [self.table[i] for i in ids if 0 <= i < len(self.table)]


_SYMBOLS = _Inventory(symbols)
_PHONEMES = _Inventory(phonemes)


# ---------------------------------------------------------------- cleaners (utils.py:558-627)
_ABBREVIATIONS = [(re.compile(r"\b%s\." % short, re.IGNORECASE), full) for short, full in (
    ("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"),
    ("jr", "junior"), ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"),
    ("lt", "lieutenant"), ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"),
    ("esq", "esquire"), ("ltd", "limited"), ("col", "colonel"), ("ft", "fort"))]


def expand_abbreviations(text):
    for pattern, full in _ABBREVIATIONS:
        text = pattern.sub(full, text)
    return text


def lowercase(text):
    return text.lower()


def collapse_whitespace(text):
    return re.sub(r"\s+", " ", text).strip()


def convert_to_ascii(text):
    try:
        from unidecode import unidecode
    except ImportError:
        if text.isascii():
            return text
        raise RuntimeError("convert_to_ascii: the `unidecode` package is needed for non-ASCII text")
    return unidecode(text)


def expand_numbers(text):
    return normalize_numbers(text)


def basic_cleaners(text):
    return collapse_whitespace(lowercase(text))


def transliteration_cleaners(text):
    return collapse_whitespace(lowercase(convert_to_ascii(text)))


def english_cleaners(text):
    return collapse_whitespace(expand_abbreviations(expand_numbers(lowercase(convert_to_ascii(text)))))


def phoneme_cleaners(text):
    return collapse_whitespace(expand_abbreviations(expand_numbers(convert_to_ascii(text))))


_CLEANERS = {f.__name__: f for f in (expand_abbreviations, expand_numbers, lowercase, collapse_whitespace,
                                     convert_to_ascii, basic_cleaners, transliteration_cleaners,
                                     english_cleaners, phoneme_cleaners)}


def _clean_text(text, cleaner_names):
    for name in cleaner_names:
        if name not in _CLEANERS:
            raise Exception("Unknown cleaner: %s" % name)
        text = _CLEANERS[name](text)
    return text


# ---------------------------------------------------------------- numbers (utils.py:629-743)
_SMALL = ("", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven",
          "twelve", "thirteen", "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen")
_TENS = ("", "ten", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety")
_GROUPS = ("", "thousand", "million", "billion", "trillion", "quadrillion")
_ORDINALS = (("one", "first"), ("two", "second"), ("three", "third"), ("five", "fifth"), ("eight", "eighth"),
             ("nine", "ninth"), ("twelve", "twelfth"), ("ty", "tieth"))


def _spell_groups(n, group):
    """n in words, thousands group by thousands group; `group` names the lowest one."""
    words = []
    if n >= 1000:
        words.append(_spell_groups(n // 1000, group + 1))
        n %= 1000
    if n >= 100:
        words.append(_SMALL[n // 100] + " hundred")
    low = n % 100
    if low >= len(_SMALL):
        words += [_TENS[low // 10], _SMALL[low % 10]]
    else:
        words.append(_SMALL[low])
    if n > 0:
        words.append(_GROUPS[group])
    return " ".join(w for w in words if w)


def _spell(n):
    if n >= 10 ** 18:
        return str(n)
    if n == 0:
        return "zero"
    if n % 100 == 0 and n % 1000 != 0 and n < 3000:      # "nineteen hundred"
        return _spell_groups(n // 100, 0) + " hundred"
    return _spell_groups(n, 0)


def _spell_ordinal(match):
    words = _spell(int(match.group(1)))
    for ending, ordinal in _ORDINALS:
        if words.endswith(ending):
            return words[:-len(ending)] + ordinal
    return words + "th"


def _spell_dollars(match):
    amount = match.group(1)
    parts = amount.split(".")
    if len(parts) > 2:
        return amount + " dollars"
    dollars = int(parts[0]) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    said = []
    if dollars:
        said.append("%s %s" % (dollars, "dollar" if dollars == 1 else "dollars"))
    if cents:
        said.append("%s %s" % (cents, "cent" if cents == 1 else "cents"))
    return ", ".join(said) if said else "zero dollars"


def normalize_numbers(text):
    text = re.sub(r"([0-9][0-9\,]+[0-9])", lambda m: m.group(1).replace(",", ""), text)
    text = re.sub(r"£([0-9\,]*[0-9]+)", r"\1 pounds", text)
    text = re.sub(r"\$([0-9\.\,]*[0-9]+)", _spell_dollars, text)
    text = re.sub(r"([0-9]+\.[0-9]+)", lambda m: m.group(1).replace(".", " point "), text)
    text = re.sub(r"([0-9]+)(st|nd|rd|th)", _spell_ordinal, text)
    text = re.sub(r"[0-9]+", lambda m: _spell(int(m.group(0))), text)
    return text


# ---------------------------------------------------------------- text <-> ids (utils.py:470-556)
_BRACES = re.compile(r"(.*?)\{(.+?)\}(.*)")


def text_to_sequence(text, cleaner_names):
    """Character ids; `{...}` holds space-separated phoneme symbols that bypass the cleaners."""
    ids = []
    while text:
        m = _BRACES.match(text)
        if m is None:
            ids += _SYMBOLS.encode(_clean_text(text, cleaner_names))
            break
        ids += _SYMBOLS.encode(_clean_text(m.group(1), cleaner_names))
        ids += _SYMBOLS.encode("@" + s for s in m.group(2).split())
        text = m.group(3)
    return ids


def sequence_to_text(sequence):
    out = "".join("{%s}" % s[1:] if len(s) > 1 and s[0] == "@" else s for s in _SYMBOLS.decode(sequence))
    return out.replace("}{", " ")


def sequence_to_phoneme(sequence):
    return "".join(_PHONEMES.decode(sequence)).replace("}{", " ")


def pad_with_eos_bos(phoneme_sequence):
    return [_PHONEMES.index[BOS]] + list(phoneme_sequence) + [_PHONEMES.index[EOS]]


def phoneme_string_to_sequence(ph, enable_eos_bos=False):
    """Ids of a phonemised string in the reference's "|"-separated form (text2phone's output)."""
    ids = []
    for chunk in filter(None, ph.split("|")):
        ids += _PHONEMES.encode(chunk)
    return pad_with_eos_bos(ids) if enable_eos_bos else ids


def restore_punctuation(text, ph):
    """The part of text2phone after the phonemizer call (utils.py:452-468): espeak drops sentence
    punctuation and leaves a line break; put the punctuation marks back in order."""
    marks = re.findall("[" + _PHONEME_PUNCT + "]+", text)
    ph = ph[:-1].strip()
    if marks:
        ends_with_mark = text[-1] == marks[-1]
        for mark in (marks[:-1] if ends_with_mark else marks):
            ph = ph.replace("| |\n", "|" + mark + "| |", 1)
        if ends_with_mark:
            ph += marks[-1]
    return ph


def text2phone(text, language, g2p=None):
    if g2p is None:
        try:
            from phonemizer.phonemize import phonemize
            from phonemizer.separator import Separator
        except ImportError:
            raise RuntimeError("text2phone needs the `phonemizer` package with espeak, or g2p=")

        def g2p(t, lang):
            return phonemize(t, separator=Separator(" |", "", "|"), strip=False, njobs=1, backend="espeak",
                             language=lang)
    return restore_punctuation(text, g2p(text, language))


def phoneme_to_sequence(text, cleaner_names, language, enable_eos_bos=False, g2p=None):
    clean = _clean_text(text.replace(":", ""), cleaner_names)
    return phoneme_string_to_sequence(text2phone(clean, language, g2p), enable_eos_bos)
