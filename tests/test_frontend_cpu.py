"""CPU: the batch producer (collate, lexicon G2P, text -> batch) against what the reference's own
``TTSDataset._collate_fn`` / ``EnglishG2P`` returned for the same inputs (tools/gen_golden_frontend.py)."""
import json
import os

import numpy as np
import pytest
import torch

from lightningfastspeech2_amd.frontend import LexiconG2P, collate, flatten, text_to_batch

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_collate.npz"))


def samples(seed):  # same recipe as tools/gen_golden_frontend.py
    rs = np.random.RandomState(seed)
    out = []
    for L, T in [(7, 31), (11, 50), (4, 18)]:
        out.append({"id": f"utt{L}", "text": "x" * L, "phones": rs.randint(1, 40, L).astype(np.int64),
                    "mel": rs.standard_normal((T, 80)).astype(np.float32), "duration": rs.randint(1, 6, L).astype(np.int64),
                    "speaker": rs.standard_normal(256).astype(np.float32),
                    "variances": {"pitch": rs.standard_normal(T).astype(np.float32),
                                  "energy": rs.standard_normal(T).astype(np.float32)},
                    "priors": {"pitch": float(rs.standard_normal())},
                    "silence_mask": (rs.rand(T) > 0.7)})
    return out


@pytest.mark.parametrize("tag,mult", [("plain", None), ("mult8", 8)])
def test_collate_matches_reference(tag, mult):
    got = collate(samples(5), pad_to_multiple_of=mult)
    keys = json.loads(str(Z[f"{tag}__keys"]))
    assert list(got.keys()) == keys            # same keys, same order (dict order is part of the contract)
    for k in keys:
        ref = Z[f"{tag}__{k}"]
        if torch.is_tensor(got[k]):
            assert got[k].dtype == torch.from_numpy(ref).dtype, k
            assert np.array_equal(got[k].numpy(), ref), k
        else:
            want = json.loads(str(ref))
            have = [x.tolist() if isinstance(x, np.ndarray) else x for x in got[k]]
            assert have == want, k


def test_collate_feeds_the_forward_contract():
    b = collate(samples(5))
    assert b["phones"].dtype == torch.int64 and b["phones"].shape == (3, 11) and int(b["phones"][2, 4:].abs().sum()) == 0
    assert b["speaker"].shape == (3, 256) and b["phones_lengths"].tolist() == [7, 11, 4]
    assert b["silence_mask"][2, 18:].all()     # silence masks pad with 1 (datasets.py:868)
    assert flatten({"a": {"b": {"c": 1}}, "d": 2}) == {"a_b_c": 1, "d": 2}


def test_lexicon_g2p_matches_reference():
    lex = json.loads(str(Z["g2p_lexicon"]))
    g = LexiconG2P(lexicon=lex)
    for text, want in zip(json.loads(str(Z["g2p_texts"])), json.loads(str(Z["g2p_out"]))):
        assert g(text) == want
    with pytest.raises(KeyError):
        g("hello unknownword")
    with pytest.raises(IndexError):            # the reference's word[-1] on a doubled space (g2p.py:34)
        g("hello  world")
    assert LexiconG2P(lexicon=lex, fallback=lambda w: ["?"])("hello zzz") == lex["hello"] + ["[SILENCE]", "?", "[SILENCE]"]


def test_arpabet_g2p_matches_reference_call():
    """Stress-digit stripping + ARPAbet -> IPA conversion: the fixture is what the reference's own EnglishG2P.__call__
    returned with a NON-identity converter (the table) and ARPAbet lexicon / fallback phones carrying stress digits."""
    from lightningfastspeech2_amd.frontend import ARPABET_TO_IPA, ArpabetConverter
    lex, oov = json.loads(str(Z["g2p2_lexicon"])), json.loads(str(Z["g2p2_oov"]))
    g = LexiconG2P(lexicon=lex, fallback=lambda w: oov[w])
    for text, want in zip(json.loads(str(Z["g2p2_texts"])), json.loads(str(Z["g2p2_out"]))):
        assert g(text) == want
    out = g("hello everything")
    assert out[:4] == ["h", "ʌ", "l", "oʊ"]           # AH0 -> AH -> ʌ, OW1 -> oʊ: digits gone before the table
    assert "ɪ" in out and "IH2" not in out             # the '2' the reference leaves on is tolerated by the table
    assert len(ARPABET_TO_IPA) == 40 and ArpabetConverter()("ZH") == ["ʒ"] and ArpabetConverter()("ə") == ["ə"]
    with pytest.raises(ValueError):
        ArpabetConverter()("a", "xsampa")


def test_lexicon_file_and_text_to_batch(tmp_path):
    p = tmp_path / "lex.txt"
    p.write_text("hello\th ə l oʊ\n\nworld\tw ɜː l d\n", encoding="utf-8")
    g = LexiconG2P(lexicon_path=str(p))
    phone2id = {"[PAD]": 0, "h": 1, "ə": 2, "l": 3, "oʊ": 4, "w": 5, "d": 6, "[SILENCE]": 7, "[FULL STOP]": 8}
    b = text_to_batch(phone2id, g, "Hello world.", np.ones(256, np.float32))
    # 'ɜː' is not in phone2id and is dropped, like generator.py:97-99
    assert b["phones"].tolist() == [[1, 2, 3, 4, 7, 5, 3, 6, 8]]
    assert b["speaker"].shape == (1, 256) and b["speaker"].dtype == torch.float32
