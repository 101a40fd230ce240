"""The batch-dict producer in front of the mel forward — host code, no GPU.

* ``collate`` restates ``TTSDataset._collate_fn`` (litfass/dataset/datasets.py:852-882): nested sample
  dicts flattened with ``_`` (``variances`` -> ``variances_pitch`` ...), numpy -> tensors, a
  ``<key>_lengths`` tensor for every array key, zero padding to the longest (``silence_mask`` keys pad
  with 1), and the ``pad_to_multiple_of`` quirk kept as is: only the FIRST sample is padded to the
  rounded-up length (``pad_sequence`` then stretches the rest).
* ``LexiconG2P`` is ``EnglishG2P.__call__`` (litfass/synthesis/g2p.py:22-65): NFKD + lower case, split on
  spaces, trailing ``. , ! ?`` become ``[<unicode name>]`` tokens, every other word boundary ``[SILENCE]``;
  every phone of a word - lexicon entry or fallback output - loses its ``0`` / ``1`` stress digits (``2``
  stays, g2p.py:44) and goes through the ARPAbet -> IPA converter, whose result EXTENDS the phone list
  (g2p.py:45-46).  The reference's converter is the third-party ``phones`` package (pyproject.toml:15,
  ``phones>=0.0.2``, not in the checkout, not in this image): ``ArpabetConverter`` restates the published
  ARPAbet -> IPA correspondence (CMUdict's 39 phonemes) as a table.  Its out-of-lexicon fallback is the
  neural ``g2p_en`` model (pyproject.toml:31), also absent: such words go to an optional ``fallback``
  callable (ARPAbet with stress digits, what ``g2p_en.G2p`` returns) or raise.
* ``text_to_batch`` is the tensor-building part of ``SpeechGenerator.generate_from_text``
  (litfass/synthesis/generator.py:96-150).
"""
from __future__ import annotations

import unicodedata
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence


def flatten(structure, key: str = "", path: str = "", flattened: Optional[dict] = None) -> dict:
    """TTSDataset._flatten (datasets.py:839-850)."""
    if flattened is None:
        flattened = {}
    if not isinstance(structure, dict):
        flattened[((path + "_") if path else "") + key] = structure
    else:
        for new_key, value in structure.items():
            flatten(value, new_key, ((path + "_") if path else "") + key, flattened)
    return flattened


def collate(samples: List[dict], pad_to_multiple_of: Optional[int] = None, load_stats_only: bool = False) -> dict:
    """TTSDataset._collate_fn (datasets.py:852-882)."""
    flat = [flatten(x) for x in samples]
    data = {k: [d[k] for d in flat] for k in flat[0]}
    add_keys = {}
    for key in data.keys():
        if "silence_mask" in key and pad_to_multiple_of is not None:
            continue
        if isinstance(data[key][0], np.ndarray):
            data[key] = [torch.tensor(x) for x in data[key]]
            add_keys[f"{key}_lengths"] = torch.tensor([x.shape[0] for x in data[key]])
        if torch.is_tensor(data[key][0]):
            pad_val = 1 if "silence_mask" in key else 0
            add_keys[f"{key}_lengths"] = torch.tensor([x.shape[0] for x in data[key]])
            if pad_to_multiple_of is not None and (key in ["mel", "phones"] or "variances" in key or "duration" in key
                                                   or load_stats_only):
                longest = int(max(add_keys[f"{key}_lengths"]))
                max_len = int(np.ceil(longest / pad_to_multiple_of) * pad_to_multiple_of)
                first = data[key][0]
                if first.dim() == 1:
                    data[key][0] = torch.nn.functional.pad(first, (0, max_len - first.shape[0]), value=pad_val)
                elif first.dim() == 2:
                    data[key][0] = torch.nn.functional.pad(first, (0, 0, 0, max_len - first.shape[0]), value=pad_val)
            data[key] = pad_sequence(data[key], batch_first=True, padding_value=pad_val)
    data.update(add_keys)
    return data


# ARPAbet (CMUdict, stress digits removed) -> IPA.  One IPA symbol string per phoneme; diphthongs and affricates stay one
# phone (the lexica the reference is used with list "oʊ", "tʃ" as single entries).
ARPABET_TO_IPA = {
    "AA": "ɑ", "AE": "æ", "AH": "ʌ", "AO": "ɔ", "AW": "aʊ", "AX": "ə", "AY": "aɪ", "B": "b", "CH": "tʃ", "D": "d",
    "DH": "ð", "EH": "ɛ", "ER": "ɝ", "EY": "eɪ", "F": "f", "G": "ɡ", "HH": "h", "IH": "ɪ", "IY": "i", "JH": "dʒ",
    "K": "k", "L": "l", "M": "m", "N": "n", "NG": "ŋ", "OW": "oʊ", "OY": "ɔɪ", "P": "p", "R": "ɹ", "S": "s", "SH": "ʃ",
    "T": "t", "TH": "θ", "UH": "ʊ", "UW": "u", "V": "v", "W": "w", "Y": "j", "Z": "z", "ZH": "ʒ",
}


class ArpabetConverter:
    """Table-driven stand-in for ``phones.convert.Converter()(phone, "arpabet", lang=None)`` (g2p.py:45): returns the
    LIST of IPA phones of one ARPAbet symbol.  A secondary-stress digit the caller left on (``AH2``: g2p.py:44 strips only
    0 and 1) is ignored; a symbol outside the table (a lexicon that already holds IPA, punctuation) passes through."""

    def __call__(self, phone: str, source: str = "arpabet", lang=None) -> List[str]:
        if source != "arpabet":
            raise ValueError("only the ARPAbet -> IPA direction is on the synthesis path (g2p.py:45)")
        key = phone.upper().rstrip("2")
        return [ARPABET_TO_IPA.get(key, phone)]


class LexiconG2P:
    """``EnglishG2P.__call__`` (g2p.py:28-52).  Lexicon file: one ``word<TAB>p1 p2 ...`` per line (g2p.py:54-65); or
    pass a dict.  ``converter(phone, "arpabet", lang=None) -> list`` defaults to :class:`ArpabetConverter`."""

    PUNCTUATION = [".", ",", "!", "?"]

    def __init__(self, lexicon_path: Optional[str] = None, lexicon: Optional[Dict[str, List[str]]] = None,
                 fallback: Optional[Callable[[str], Iterable[str]]] = None, converter: Optional[Callable] = None):
        self.lexicon_path = lexicon_path
        self.lexicon = {k.lower(): list(v) for k, v in (lexicon or {}).items()}
        self.lexicon.update(self.load_lexicon())
        self.fallback = fallback
        self.converter = converter if converter is not None else ArpabetConverter()

    def load_lexicon(self) -> Dict[str, List[str]]:
        lex = {}
        if self.lexicon_path is not None:
            with open(self.lexicon_path, "r", encoding="utf-8") as f:
                for line in f:
                    line = line.strip()
                    if len(line) == 0:
                        continue
                    word, phones = line.split("\t")
                    lex[word.lower()] = phones.split(" ")
        return lex

    def __call__(self, text: str) -> List[str]:
        text = unicodedata.normalize("NFKD", text).lower()
        phones: List[str] = []
        for word in text.split(" "):
            if word == "":
                # the reference indexes word[-1] and raises IndexError on doubled spaces (g2p.py:34)
                raise IndexError("empty word (doubled space) in text")
            punctuation = ""
            if word[-1] in self.PUNCTUATION:
                punctuation, word = word[-1], word[:-1]
            if word in self.lexicon:
                add_phones = self.lexicon[word]
            elif self.fallback is not None:
                add_phones = list(self.fallback(word))
            else:
                raise KeyError(f"'{word}' is not in the lexicon and no fallback G2P is installed "
                               "(the reference uses the third-party g2p_en model here)")
            for phone in add_phones:  # g2p.py:42-46
                phone = phone.replace("0", "").replace("1", "")
                phones += self.converter(phone, "arpabet", lang=None)
            phones.append("[" + unicodedata.name(punctuation) + "]" if punctuation else "[SILENCE]")
        return phones


def text_to_batch(phone2id: Dict[str, int], g2p: Callable[[str], List[str]], text: str, dvector, device="cpu") -> dict:
    """``generate_from_text`` up to the model call (generator.py:96-99,118-121,146-150): phones the model
    does not know are dropped, batch of one."""
    ids = [phone2id[x] for x in g2p(text) if x in phone2id]
    return {"phones": torch.tensor([ids], dtype=torch.long, device=device),
            "speaker": torch.tensor(np.asarray([dvector]), dtype=torch.float32, device=device)}
