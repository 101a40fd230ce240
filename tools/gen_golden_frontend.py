#!/usr/bin/env python
"""tests/golden/frontend_collate.npz: what the REAL ``TTSDataset._collate_fn`` and ``EnglishG2P`` (lexicon
path) of the reference return for a few synthetic samples (build container only; the reference module is
imported with stub modules for its absent third-party dependencies — the two functions exercised use
none of them)."""
import importlib
import importlib.util
import json
import os
import sys
import types
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")


class Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return MagicMock()


def stub(name):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            m = Stub(n)
            m.__path__ = []
            sys.modules[n] = m


for n in ["pyworld", "seaborn", "torchaudio", "torchaudio.transforms", "librosa", "librosa.filters", "pandarallel", "phones",
          "phones.convert", "srmrpy", "tqdm.rich", "PIL", "matplotlib", "matplotlib.pyplot", "matplotlib.gridspec",
          "litfass.dataset.cwt", "litfass.dataset.snr", "litfass.third_party.dvectors.wav2mel", "litfass.third_party.dvectors",
          "pytorch_lightning", "wandb", "rich", "g2p_en"]:
    try:
        importlib.import_module(n)
    except Exception:
        for k in [k for k in sys.modules if k == n or k.startswith(n + ".")]:
            del sys.modules[k]
        stub(n)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def samples(seed):
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


def main():
    ds = load("/root/reference/litfass/dataset/datasets.py", "ref_datasets")
    fix = {}
    for tag, mult in (("plain", None), ("mult8", 8)):
        res = ds.TTSDataset._collate_fn(SimpleNamespace(pad_to_multiple_of=mult, _load_stats_only=False), samples(5))
        keys = []
        for k, v in res.items():
            if torch.is_tensor(v):
                fix[f"{tag}__{k}"] = v.numpy()
                keys.append(k)
            else:
                fix[f"{tag}__{k}"] = np.asarray(json.dumps([x if not isinstance(x, np.ndarray) else x.tolist() for x in v]))
                keys.append(k)
        fix[f"{tag}__keys"] = np.asarray(json.dumps(keys))
    # EnglishG2P with a lexicon that covers the text (g2p_en / phones stubbed, never reached)
    g2p = load("/root/reference/litfass/synthesis/g2p.py", "ref_g2p")
    lex = {"hello": ["h", "ə", "l", "oʊ"], "world": ["w", "ɜː", "l", "d"], "again": ["ə", "ɡ", "ɛ", "n"]}
    lex_path = "/tmp/_lexicon.txt"
    with open(lex_path, "w", encoding="utf-8") as f:
        for w, p in lex.items():
            f.write(w + "\t" + " ".join(p) + "\n")
    e = g2p.EnglishG2P.__new__(g2p.EnglishG2P)
    e.lexicon_path = lex_path
    e.lexicon = g2p.EnglishG2P.load_lexicon(e)
    e.g2p = None
    e.converter = lambda phone, *_a, **_k: [phone]   # lexicon entries are IPA already; identity conversion
    texts = ["Hello world.", "hello, world! again", "WORLD hello?"]
    fix["g2p_lexicon"] = np.asarray(json.dumps(lex))
    fix["g2p_texts"] = np.asarray(json.dumps(texts))
    fix["g2p_out"] = np.asarray(json.dumps([e(t) for t in texts]))
    # EnglishG2P on ARPAbet: lexicon entries AND the (stubbed) g2p_en fallback carry stress digits; the converter is the
    # table-driven ARPAbet -> IPA map (NOT an identity) - what the reference's own __call__ does with them is the fixture:
    # 0 / 1 stripped, 2 kept (g2p.py:44), the converter's list EXTENDS the phone list (g2p.py:45-46)
    from lightningfastspeech2_amd.frontend import ArpabetConverter
    lex2 = {"hello": ["HH", "AH0", "L", "OW1"], "world": ["W", "ER1", "L", "D"], "choice": ["CH", "OY1", "S"]}
    oov = {"thinking": ["TH", "IH1", "NG", "K", "IH0", "NG"], "about": ["AH0", "B", "AW1", "T"],
           "everything": ["EH1", "V", "R", "IY0", "TH", "IH2", "NG"], "measure": ["M", "EH1", "ZH", "ER0"]}
    with open(lex_path, "w", encoding="utf-8") as f:
        for w, p in lex2.items():
            f.write(w + "\t" + " ".join(p) + "\n")
    e2 = g2p.EnglishG2P.__new__(g2p.EnglishG2P)
    e2.lexicon_path = lex_path
    e2.lexicon = g2p.EnglishG2P.load_lexicon(e2)
    e2.g2p = lambda word: list(oov[word])          # stands in for g2p_en.G2p()(word): ARPAbet with stress digits
    e2.converter = ArpabetConverter()
    texts2 = ["Hello world, thinking about everything!", "choice measure. HELLO?", "about world"]
    fix["g2p2_lexicon"] = np.asarray(json.dumps(lex2))
    fix["g2p2_oov"] = np.asarray(json.dumps(oov))
    fix["g2p2_texts"] = np.asarray(json.dumps(texts2))
    fix["g2p2_out"] = np.asarray(json.dumps([e2(t) for t in texts2]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frontend_collate.npz"), **fix)
    print({k: (v.shape if v.ndim else "json") for k, v in fix.items()})


if __name__ == "__main__":
    main()
