#!/usr/bin/env python3
"""Golden PHOS / PHOC / PHOSC vectors from the REFERENCE's own generators (TEST INFRASTRUCTURE, run once where the
reference checkout is mounted):

    python oracle/make_golden_phosc.py /root/reference tests/golden

``ResPhoSCNetZSL/modules/utils/phos_generator.py`` and ``phoc_generator.py`` are imported unmodified (by file path; the
package ``__init__`` pulls in unrelated modules).  ``set_phos_version`` hard-codes ``/cluster/...`` csv paths, so its four
assignments are replayed here with the csv that ships next to the module.  Only data is written: the shape-count tables
(``Alphabet*.csv`` parsed exactly as the reference parses them), the words, and the reference's output vectors."""
import importlib.util
import os
import sys

import numpy as np


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref, outdir = sys.argv[1], sys.argv[2]
    ud = os.path.join(ref, "ResPhoSCNetZSL", "modules", "utils")
    phos = _load(os.path.join(ud, "phos_generator.py"), "ref_phos_generator")
    phoc = _load(os.path.join(ud, "phoc_generator.py"), "ref_phoc_generator")
    words = ["a", "to", "MOVE", "text", "getting", "prop", "Zebra", "handwritten", "Q", "of", "the", "Stop", "abcdefghij",
             "it", "we11", "x2"]
    payload = {}
    for version, csv_name in (("eng", "Alphabet.csv"), ("gw", "AlphabetGW.csv"), ("nor", "AlphabetNorwegian.csv")):
        csv_path = os.path.join(ud, csv_name)
        # == set_phos_version(version) with the csv next to the module (phos_generator.py:36-56)
        phos.alphabet_csv = csv_path
        phos.alphabet_dict = phos.create_alphabet_dictionary(csv_path)
        phos.csv_num_cols = phos.get_number_of_columns(csv_path)
        table = np.genfromtxt(csv_path, dtype=int, delimiter=",")
        phos.numpy_csv = np.delete(table, 0, 1)
        phoc.set_phoc_version(version)
        import csv
        with open(csv_path) as f:  # one entry per csv row (a letter may occur twice: the dictionary keeps the last row)
            letters = [line[0] for line in csv.reader(f, delimiter=',', skipinitialspace=True)]
        ok_words = [w for w in words if all(ch in phos.alphabet_dict for ch in w)]
        payload[f"{version}:letters"] = np.array(letters)
        payload[f"{version}:table"] = phos.numpy_csv.astype(np.int64)
        payload[f"{version}:words"] = np.array(ok_words)
        payload[f"{version}:phos"] = np.stack([phos.generate_label(w) for w in ok_words]).astype(np.float64)
        payload[f"{version}:phoc"] = np.stack([np.array(phoc.generate_phoc_vector(w)) for w in ok_words]).astype(np.int64)
        print(f"[golden] phosc {version}: {len(letters)} letters, {len(ok_words)} words, phos {payload[f'{version}:phos'].shape[1]}"
              f" + phoc {payload[f'{version}:phoc'].shape[1]}")
    np.savez_compressed(os.path.join(outdir, "phosc.npz"), **payload)


if __name__ == "__main__":
    main()
