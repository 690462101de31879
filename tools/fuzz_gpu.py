#!/usr/bin/env python3
"""Longer fuzz run of the GPU path against the oracle (the test suite runs a few seeds of the same generators).
usage: python tools/fuzz_gpu.py [decode_seeds] [compress_seeds]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests import _oracle
from tests import test_gpu_parity as T

oracle = _oracle.load()
wah = importlib.import_module("gpu-wah_amd")
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
for seed in range(nd):
    rng = np.random.default_rng(50000 + seed)
    n_words = int(rng.choice([1, 2, 63, 64, 65, 500, 4095, 4096, 4097, 8191, 8192, 8193, 4096 * 5 + 7, 4096 * 33, 4096 * 64 + 1, 4096 * 130]))
    st = T._random_foreign_stream(rng, n_words, max_groups=int(rng.choice([1000, 1024, 1025, 100_000, 12_000_000])))
    want = oracle.decompress(st)
    got = T._host(wah.decompress_device(T._dev(st), len(want) + 2))
    if len(got) != len(want) or not np.array_equal(got, want):
        bad += 1
        d = np.nonzero(got[: min(len(got), len(want))] != want[: min(len(got), len(want))])[0]
        print("DECODE MISMATCH seed", seed, "words", len(st), "len", len(got), len(want), "first diffs", d[:5])
    if len(st) <= 20000:  # the word-by-word Python restatements are slow
        if tuple(wah.validate_device(T._dev(st))) != T._py_report(st):
            bad += 1
            print("VALIDATE MISMATCH seed", seed)
        if not np.array_equal(T._host(wah.merge_fills_device(T._dev(st))), T._py_merge_fills(st)):
            bad += 1
            print("MERGE MISMATCH seed", seed)
    if seed % 20 == 0:
        print("decode seed", seed, "ok so far, bad =", bad, f"{time.time() - t0:.0f}s", flush=True)
for seed in range(nc):
    rng = np.random.default_rng(90000 + seed)
    n = int(rng.choice([1, 31, 991, 992, 993, 992 * 15, 992 * 15 + 1, 992 * 15 * 256 + 17, 992 * 4000]))
    mode = seed % 4
    if mode == 0:
        data = oracle.gen_uniform(n, seed, float(rng.choice([0.5, 0.1, 0.01, 1e-4, 0.999])))
    elif mode == 1:
        data = oracle.gen_clustered(n, seed, int(rng.choice([40, 600, 4096, 100000])))
    else:
        data = oracle.gen_uniform(n, seed, 0.3)
        data[rng.random(n) < 0.7] = 0
        data[rng.random(n) < 0.3] = 0xFFFFFFFF
    want = oracle.compress(data)
    got = T._host(wah.compress_device(T._dev(data)))
    if got.shape != want.shape or not np.array_equal(got, want):
        bad += 1
        print("COMPRESS MISMATCH seed", seed, "n", n, "mode", mode)
    back = T._host(wah.decompress_device(T._dev(want), n + 1))
    if not np.array_equal(back[:n], data):
        bad += 1
        print("ROUND TRIP MISMATCH seed", seed, "n", n, "mode", mode)
    # index decode: whole bitmap and a random range through the compressor's segment index
    stream, offs = T._indexed_stream(wah, T._dev(data))
    full = oracle.decompress(want)
    segs = (wah.max_compressed_words(n) + 1023) // 1024
    first = int(rng.integers(0, segs))
    count = int(rng.integers(0, segs - first + 1))
    if not np.array_equal(T._host(stream), want) or \
            not np.array_equal(T._host(wah.decompress_segments_device(stream, offs, n)), full) or \
            not np.array_equal(T._host(wah.decompress_segments_device(stream, offs, n, first, count)), full[first * 992: (first + count) * 992]):
        bad += 1
        print("INDEX DECODE MISMATCH seed", seed, "n", n, "mode", mode, first, count)
    bo, bg = wah.build_index_device(T._dev(want))  # the index rebuilt from the bare stream
    if bg != wah.max_compressed_words(n) or not np.array_equal(bo.cpu().numpy(), offs.cpu().numpy()):
        bad += 1
        print("BUILD INDEX MISMATCH seed", seed, "n", n, "mode", mode)
    if n >= 992 and seed % 3 == 0:  # bitwise operations against numpy on the decoded bitmaps
        other = oracle.gen_clustered(n, seed + 7, 900)
        for name, fn in (("and", np.bitwise_and), ("or", np.bitwise_or), ("xor", np.bitwise_xor), ("andnot", lambda x, y: x & ~y)):
            w = oracle.compress(fn(data, other).astype(np.uint32))
            g = T._host(wah.bitop_device(name, T._dev(want), T._dev(oracle.compress(other)), n))
            if g.shape != w.shape or not np.array_equal(g, w):
                bad += 1
                print("BITOP MISMATCH seed", seed, name, "n", n)
            so, oo = T._indexed_stream(wah, T._dev(other))
            gi, _ = wah.bitop_indexed_device(name, stream, offs, so, oo, n)
            if gi.numel() != w.size or not np.array_equal(T._host(gi), w):
                bad += 1
                print("INDEXED BITOP MISMATCH seed", seed, name, "n", n)
    if seed % 10 == 0:
        print("compress seed", seed, "bad =", bad, f"{time.time() - t0:.0f}s", flush=True)
print("done, mismatches:", bad)
sys.exit(1 if bad else 0)
