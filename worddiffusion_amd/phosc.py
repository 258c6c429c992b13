"""PHOS / PHOC / PHOSC word descriptors on the host: the integer vector ``UNetModelPhosc`` takes as
``phoscLabels`` ([B, 769] for English: 165 PHOS + 604 PHOC).

Restated from the reference's ``ResPhoSCNetZSL/modules/utils/phos_generator.py:60-78`` (pyramid of shape counts: the whole
word, then its 2, 3, 4, 5 way splits - 15 segments x the csv's shape columns) and ``phoc_generator.py:17-90`` (pyramid of
character-presence vectors over the 2..5 way splits + two 50-entry bigram blocks), combined as in
``datasets.py:44-70`` / ``trainGWModifyCondition.py:394-409`` (blanks and underscores removed, ``phos ++ phoc``, int64).
Behaviours of the reference that are kept because the checkpoints were trained with them:
  * the bigram blocks are always zero: ``generate_50`` looks single characters up in a list of two-letter strings
    (``phoc_generator.py:69-74``);
  * version ``'gw'`` has no branch in ``generate_chars`` (``:28-47``), so its whole PHOC is zero;
  * PHOC lower-cases the word, PHOS does not (``:81`` vs ``phos_generator.py:60-66``); unknown letters raise ``KeyError``.
The shape-count table is the reference's data file (``Alphabet.csv`` / ``AlphabetGW.csv`` / ``AlphabetNorwegian.csv``): pass
its path (``load_alphabet``) - it is not shipped here.  Golden vectors: ``tests/golden/phosc.npz``."""
from __future__ import annotations

import csv
from typing import Dict, List, Sequence, Tuple

import numpy as np

BIGRAMS = {
    "eng": ['th', 'he', 'in', 'er', 'an', 're', 'es', 'on', 'st', 'nt', 'en', 'at', 'ed', 'nd', 'to', 'or', 'ea', 'ti', 'ar',
            'te', 'ng', 'al', 'it', 'as', 'is', 'ha', 'et', 'se', 'ou', 'of', 'le', 'sa', 've', 'ro', 'ra', 'hi', 'ne', 'me',
            'de', 'co', 'ta', 'ec', 'si', 'll', 'so', 'na', 'li', 'la', 'el', 'ma'],
    "nor": ['de', 'og', 'ha', 'je', 'at', 'me', 'fo', 'en', 'ti', 'er', 'mi', 'vi', 'so', 'sa', 'he', 'si', 'ik', 'af', 'sk',
            'st', 'ma', 'be', 'hv', 'al', 'fr', 'va', 've', 'om', 'pa', 'et', 'se', 'di', 'da', 'li', 'bl', 'in', 'du', 'no',
            'ko', 'an', 'væ', 'fa', 'ku', 'ka', 'ga', 'hu', 'ta', 're', 'ud', 'op'],
}
BIGRAMS["gw"] = BIGRAMS["eng"]


def load_alphabet(csv_path: str) -> Tuple[Dict[str, int], np.ndarray]:
    """(letter -> row, int table [letters, shapes]) parsed as ``phos_generator.py:22-33,51-52`` does."""
    index = {}
    with open(csv_path) as f:
        for i, line in enumerate(csv.reader(f, delimiter=",", skipinitialspace=True)):
            index[line[0]] = i
    table = np.delete(np.genfromtxt(csv_path, dtype=int, delimiter=","), 0, 1)
    return index, table


def _segments(word: str) -> List[str]:
    """The 2..5 way splits shared by both pyramids (``phos_generator.py:70-76``, ``phoc_generator.py:83-88``)."""
    out, n = [], len(word)
    for split in range(2, 6):
        parts = n // split
        for mul in range(split - 1):
            out.append(word[mul * parts:mul * parts + parts])
        out.append(word[(split - 1) * parts:n])
    return out


def phos_vector(word: str, index: Dict[str, int], table: np.ndarray) -> np.ndarray:
    """``generate_label`` (phos_generator.py:66-78): float64 [15 * shapes]."""
    def count(seg):
        v = np.zeros(table.shape[1])
        for letter in seg:
            v += table[index[letter]]
        return v
    return np.concatenate([count(word)] + [count(s) for s in _segments(word)], axis=0)


# PHOC character axis per version (phoc_generator.py:17-47): digits, then a-z, then the three Norwegian letters; version 'gw' has
# no branch there, so nothing is ever set (36 zeros).
_PHOC_AXIS = {"eng": "0123456789abcdefghijklmnopqrstuvwxyz", "nor": "0123456789abcdefghijklmnopqrstuvwxyzæøå", "gw": ""}
_PHOC_SIZE = {"eng": 36, "nor": 39, "gw": 36}


def _chars(seg: str, version: str) -> List[int]:
    """Character-presence vector of one segment: one table lookup per character.  A digit or letter outside the version's axis is an
    index error, as in the reference (its ``ord`` offset lands past the vector); anything else is skipped."""
    axis = _PHOC_AXIS[version]
    vector = [0] * _PHOC_SIZE[version]
    for ch in seg:
        pos = axis.find(ch)
        if pos >= 0:
            vector[pos] = 1
        elif axis and (ch.isdigit() or ch.isalpha()):
            raise IndexError(f"PHOC: {ch!r} is not on the {version!r} character axis")
    return vector


def _bigram_block(seg: str, version: str) -> List[int]:
    vector = [0] * 50
    for ch in seg:  # single characters against two-letter strings: never found (the reference's behaviour)
        if ch in BIGRAMS[version]:
            vector[BIGRAMS[version].index(ch)] = 1
    return vector


def phoc_vector(word: str, version: str = "eng") -> List[int]:
    """``generate_phoc_vector`` (phoc_generator.py:78-90)."""
    if version not in BIGRAMS:
        raise ValueError(version)
    word = word.lower()
    vector: List[int] = []
    for seg in _segments(word):
        vector += _chars(seg, version)
    n = len(word)
    vector += _bigram_block(word[0:n // 2], version)
    vector += _bigram_block(word[n // 2:n], version)
    return vector


def phosc_vector(word: str, index: Dict[str, int], table: np.ndarray, version: str = "eng", phosc: int = 1,
                 phos: int = 0) -> np.ndarray:
    """int64 descriptor of one word as the dataset builds it (``datasets.py:49-67``)."""
    w = word.replace(" ", "").replace("_", "")
    ph = phos_vector(w, index, table)
    if phosc == 1:
        out = np.concatenate((ph, np.array(phoc_vector(w, version), dtype=np.float32)))
    else:
        out = ph
    return out.astype(np.int64)


def phosc_batch(words: Sequence[str], index, table, version: str = "eng"):
    import torch
    return torch.from_numpy(np.stack([phosc_vector(w, index, table, version) for w in words]))
