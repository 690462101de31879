#!/usr/bin/env python3
"""Debug helper: compress one known-answer vector on the GPU and print where it differs from the expected stream."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as oracle
wah = importlib.import_module("gpu-wah_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "merge_all"
for k in oracle.load_kats():
    if k["name"] == name:
        got = wah.compress(k["data"])
        exp = k["expected"]
        print("n_words", k["n_words"], "expected words", len(exp), "got", len(got))
        n = min(len(exp), len(got))
        diff = np.nonzero(got[:n] != exp[:n])[0]
        print("first diffs", diff[:10])
        for i in diff[:10]:
            print(i, hex(got[i]), hex(exp[i]))
        print("got", [hex(x) for x in got[:12]])
        print("exp", [hex(x) for x in exp[:12]])
