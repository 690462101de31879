#!/usr/bin/env python3
"""Debug helper: decode a few odd streams on the GPU and report the first difference against the oracle."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _oracle
oracle = _oracle.load()
wah = importlib.import_module("gpu-wah_amd")
streams = [
    np.array([0x80000000 | 5000, 0x12345, 0xC0000000 | 3000, 0x80000001, 0x7FFFFFFE], np.uint32),
    np.array([0x80000000, 0xC0000000, 7, 0x80000000 | 40, 0xC0000000, 9], np.uint32),
    np.array([3] * 5000 + [0x80000000 | 100] * 4000 + [0xC0000000 | 7] * 3000 + [0x80000000] * 500 + [5], np.uint32),
    np.concatenate([np.full(9000, 0x80000000 | 3, np.uint32), np.arange(1, 9001, dtype=np.uint32)]),
]
for i, st in enumerate(streams):
    want = oracle.decompress(st)
    d = torch.from_numpy(st.view(np.int32)).cuda()
    try:
        got = wah.decompress_device(d, len(want) + 3).cpu().numpy().view(np.uint32)
    except Exception as e:
        print(i, "error", e); continue
    n = min(len(got), len(want))
    diff = np.nonzero(got[:n] != want[:n])[0]
    print(i, "len got", len(got), "want", len(want), "diffs", len(diff), diff[:8], [hex(got[j]) for j in diff[:4]], [hex(want[j]) for j in diff[:4]])
