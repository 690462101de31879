#!/usr/bin/env python3
"""Generate tests/golden/kats.json -- the known-answer vectors for compress().

The vectors are DATA restated from the reference's tests (inputs and expected
outputs; `/root/reference/tests.cpp` line ranges are cited per entry).  No
reference code is imported, executed or copied: the reference is CUDA and
cannot run in this pipeline, so the expected words are
the ones its test file stores, written down here as numbers.

Status values
  must_pass : the stored vector is consistent with the shipped kernels.cu.
  stale     : the stored vector predates `|| counts[id] > 1` (kernels.cu:195);
              `expected` holds the canonical answer of the shipped kernel,
              `stale_expected` the vector the reference file still stores.
  derived   : implied by the format rules, no stored vector in the reference.
Entries are sparse: `input` maps word index -> value, all other words are 0.
"""
import json
import os

BIT31 = 0x80000000
BIT3130 = 0xC0000000
ONES = 0xFFFFFFFF
M31 = 0x7FFFFFFF


def warp_pattern(base):
    # tests.cpp:23-31 (generateTestData): 8 | 0 | . | 4<<28 | 0 | 63<<26 | ONES | ONES>>8
    return {base + 0: 8, base + 3: (4 << 28) & ONES, base + 5: (63 << 26) & ONES, base + 6: ONES, base + 7: ONES >> 8}


WARP_EXPECTED = [8, 3 | BIT31, 4, 1 | BIT31, 2 | BIT3130, 24 | BIT31]  # tests.cpp:146


def wandering(base):
    # tests.cpp:33-39 (generateWanderingTestData): a literal `1` at groups 33*w, w = 0..31
    d = {base: 1, base + 31: 1 << 31}
    for i in range(30):
        d[base + 31 + (i + 1) * 32] = 1 << (30 - i)
    return d


def wandering_stale_expected():
    # tests.cpp:66-77 (generateWanderingExpectedData), 93 words
    e = [0] * 93
    e[0] = 1
    e[1] = BIT31 | 31
    for i in range(30):
        e[2 + 3 * i] = BIT31 | (i + 1)
        e[2 + 3 * i + 1] = 1
        e[2 + 3 * i + 2] = BIT31 | (30 - i)
    e[91] = BIT31 | 32
    e[92] = 1
    return e


WANDERING_CANONICAL = [1] + [BIT31 | 32, 1] * 31  # shipped kernel, SURVEY appendix B


def divide_case():
    # tests.cpp:83-104: data = 1..31; expected[i] = M31 & (data[i] << i | data[i-1] >> (32-i)), data[31] := 0
    data = list(range(1, 32)) + [0]
    exp = [data[0] & M31]
    for i in range(1, 32):
        exp.append(M31 & (((data[i] << i) & ONES) | (data[i - 1] >> (32 - i))))
    return {i: v for i, v in enumerate(data[:31])}, exp


def main():
    kats = []

    inp, exp = divide_case()
    kats.append(dict(name="divide", source="tests.cpp:83-104", status="must_pass", n_words=31, input=inp,
                     expected=exp))

    kats.append(dict(name="warp", source="tests.cpp:134-152, data :23-31", status="must_pass", n_words=31,
                     input=warp_pattern(0), expected=WARP_EXPECTED))

    d = {}
    for w in range(32):
        d.update(warp_pattern(31 * w))
    kats.append(dict(name="block", source="tests.cpp:154-164", status="must_pass", n_words=992, input=d,
                     expected=WARP_EXPECTED * 32))

    kats.append(dict(name="merge_all", source="tests.cpp:166-172", status="must_pass", n_words=992, input={},
                     expected=[BIT31 | 1024]))

    kats.append(dict(name="ones_starts", source="tests.cpp:174-185", status="must_pass", n_words=992,
                     input={31 * i: ONES for i in range(0, 32, 2)}, expected=[BIT3130 | 1, 1, BIT31 | 62] * 16))

    d = {}
    for i in range(2, 32, 4):
        for j in range(62):
            d[31 * i + j] = ONES
    kats.append(dict(name="alternating", source="tests.cpp:187-199", status="must_pass", n_words=992, input=d,
                     expected=[BIT31 | 64, BIT3130 | 64] * 8))

    kats.append(dict(name="final_literals", source="tests.cpp:201-211", status="must_pass", n_words=992,
                     input={31 * (i + 1) - 1: 88 for i in range(32)}, expected=[BIT31 | 31, 44] * 32))

    kats.append(dict(name="wandering", source="tests.cpp:213-225, data :33-39, stored vector :66-77",
                     status="stale", n_words=992, input=wandering(0), expected=WANDERING_CANONICAL,
                     stale_expected=wandering_stale_expected()))

    d = wandering(0)
    d.update(wandering(992))
    kats.append(dict(name="multi_block", source="tests.cpp:227-239", status="stale", n_words=1984, input=d,
                     expected=WANDERING_CANONICAL * 2, stale_expected=wandering_stale_expected() * 2))

    kats.append(dict(name="two_zero_blocks", source="kernels.cu:188-229 (merge is intra-block), tests.cpp:166-172",
                     status="derived", n_words=1984, input={}, expected=[BIT31 | 1024] * 2))

    for k in kats:
        k["input"] = {str(i): int(v) for i, v in sorted(k["input"].items())}
        k["expected"] = [int(v) for v in k["expected"]]
        if "stale_expected" in k:
            k["stale_expected"] = [int(v) for v in k["stale_expected"]]

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")
    with open(out, "w") as f:
        json.dump(dict(format="wah31-seg1024", kats=kats), f, indent=0, separators=(",", ":"))
    print(f"wrote {out}: {len(kats)} vectors")


if __name__ == "__main__":
    main()
