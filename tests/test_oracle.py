"""CPU tests: the oracle against the reference's golden vectors and against itself.

Mirrors the reference's test strategy: known-answer tests of
compress() output words on hand-built inputs (tests.cpp:83-239) and round-trip
identity on larger inputs (tests.cpp:241-307).
"""
import numpy as np
import pytest

from tests import _oracle

M31 = 0x7FFFFFFF


def test_kats_must_pass(oracle, kats):
    n = 0
    for k in kats:
        got = oracle.compress(k["data"])
        assert np.array_equal(got, k["expected"]), k["name"]
        n += 1
    assert n == 10


def test_kats_refsim_matches_stored_vectors(oracle, kats):
    """The lane-level emulation of the shipped kernel gives the canonical words on every whole-block KAT."""
    for k in kats:
        if k["n_words"] % 992:
            continue
        assert np.array_equal(oracle.refsim_compress(k["data"]), k["expected"]), k["name"]


def test_stale_vectors_are_the_pre195_kernel(oracle, kats):
    """tests.cpp:66-77 / :227-239 store 93 / 186 words: exactly the kernel without `|| counts[id] > 1`."""
    stale = [k for k in kats if k["status"] == "stale"]
    assert len(stale) == 2
    for k in stale:
        assert len(k["stale_expected"]) in (93, 186)
        assert np.array_equal(oracle.refsim_compress_pre195(k["data"]), k["stale_expected"]), k["name"]
        assert not np.array_equal(oracle.compress(k["data"]), k["stale_expected"])
        # both encodings describe the same bitmap
        assert np.array_equal(oracle.decompress(k["stale_expected"])[: k["n_words"]], k["data"])


def test_kats_python_restatement(oracle, kats):
    for k in kats:
        assert np.array_equal(_oracle.py_compress(k["data"]), k["expected"]), k["name"]
        assert np.array_equal(_oracle.py_decompress(k["expected"])[: k["n_words"]], k["data"]), k["name"]


def test_appendix_a_groups(oracle, kats):
    """group values of the warp pattern (pins bit order and regroup, kernels.cu:79)."""
    warp = next(k for k in kats if k["name"] == "warp")["data"]
    vals = [oracle.group(warp, g) for g in range(32)]
    assert vals[0] == 8 and vals[4] == 4
    assert vals[1:4] == [0, 0, 0] and vals[5] == 0
    assert vals[6] == M31 and vals[7] == M31
    assert all(v == 0 for v in vals[8:])


def _structured_block(rng):
    """992 words made of 31-bit groups drawn as zero / ones / literal with random run structure."""
    groups = []
    while len(groups) < 1024:
        kind = rng.integers(0, 4)
        run = int(rng.choice([1, 1, 2, 3, 7, 31, 32, 33, 64, 100, 300]))
        if kind == 0:
            groups += [0] * run
        elif kind == 1:
            groups += [M31] * run
        else:
            groups += [int(x) for x in rng.integers(1, M31, size=min(run, 4))]
    groups = groups[:1024]
    bits = 0
    for i, g in enumerate(groups):
        bits |= g << (31 * i)
    return np.array([(bits >> (32 * j)) & 0xFFFFFFFF for j in range(992)], np.uint32)


def test_refsim_equals_canonical_on_structured_blocks(oracle):
    """The shipped kernel's warp-0 merge phase (kernels.cu:188-229) == maximal runs inside the block."""
    rng = np.random.default_rng(1337)
    for it in range(600):
        nblk = 1 + it % 3
        data = np.concatenate([_structured_block(rng) for _ in range(nblk)])
        a = oracle.compress(data)
        b = oracle.refsim_compress(data)
        assert np.array_equal(a, b), f"iteration {it}"


def test_refsim_equals_canonical_on_random_density(oracle):
    for i, p in enumerate([0.5, 0.1, 0.01, 0.001, 0.999, 0.9]):
        data = oracle.gen_uniform(992 * 8, 1337 + i, p)
        assert np.array_equal(oracle.compress(data), oracle.refsim_compress(data))
    data = oracle.gen_clustered(992 * 64, 7)
    assert np.array_equal(oracle.compress(data), oracle.refsim_compress(data))


def test_python_restatement_on_random(oracle):
    rng = np.random.default_rng(3)
    for n in [0, 1, 2, 30, 31, 32, 61, 62, 63, 100, 991, 992, 993, 1100]:
        data = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
        data[rng.random(n) < 0.5] = 0
        data[rng.random(n) < 0.2] = 0xFFFFFFFF
        c = oracle.compress(data)
        assert np.array_equal(c, _oracle.py_compress(data)), n
        d = oracle.decompress(c)
        assert np.array_equal(d, _oracle.py_decompress(c)), n


@pytest.mark.parametrize("n", [0, 1, 5, 30, 31, 32, 62, 93, 500, 991, 992, 993, 1984, 2000, 31 * 1000, 261888, 262144])
def test_round_trip_and_sizes(oracle, n):
    """F5 / SURVEY H9: decoded size = ceil(31*G/32): n when n % 31 == 0, else n + 1 with a zero pad word."""
    data = oracle.gen_uniform(n, 99, 0.03)
    c = oracle.compress(data)
    G = oracle.max_words(n)
    assert oracle.decoded_groups(c) == G
    d = oracle.decompress(c)
    want = n if n % 31 == 0 else n + 1
    assert len(d) == oracle.decoded_words(G) == want
    assert np.array_equal(d[:n], data)
    assert not d[n:].any()
    assert len(c) <= G


def test_format_invariants(oracle):
    """F2-F4: no literal equals 0 / 0x7FFFFFFF, fill counts in 1..1024, adjacent fills of one kind only across segments."""
    for p in (0.5, 0.01, 0.0001):
        data = oracle.gen_uniform(992 * 40 + 17, 5, p)
        c = oracle.compress(data)
        fill = (c & 0x80000000) != 0
        lit = c[~fill]
        assert not np.any(lit == 0) and not np.any(lit == M31)
        cnt = c[fill] & 0x3FFFFFFF
        if cnt.size:
            assert cnt.min() >= 1 and cnt.max() <= 1024
        # walk the stream: same-kind adjacent fills must be split exactly at a 1024-group boundary
        pos = 0
        prev_kind = None
        for w in c.tolist():
            if w & 0x80000000:
                kind = w >> 30
                if prev_kind == kind:
                    assert pos % 1024 == 0
                pos += w & 0x3FFFFFFF
                prev_kind = kind
            else:
                pos += 1
                prev_kind = None


def test_config1_dense_1mib(oracle):
    """BASELINE config 1: 1 MiB uniform p=0.5 -> all literals, C = G, ratio 32/31 (both N variants)."""
    for n in (261888, 262144):
        data = oracle.gen_uniform(n, 1337, 0.5)
        c = oracle.compress(data)
        G = oracle.max_words(n)
        # p=0.5: P(group is a fill) = 2 * 2^-31, so every group is a literal except maybe the zero-padded tail
        assert abs(len(c) - G) <= 1
        assert np.array_equal(oracle.decompress(c)[:n], data)


def test_expected_ratios(oracle):
    """C/N for the bench distributions (analytic model, re-measured here)."""
    n = 992 * 2048
    r = len(oracle.compress(oracle.gen_uniform(n, 1337, 0.01))) / n
    assert 0.46 < r < 0.50  # analytic 0.479
    r = len(oracle.compress(oracle.gen_uniform(n, 1337, 2.0**-4))) / n
    assert 1.00 < r < 1.03  # analytic 1.013
    r = len(oracle.compress(oracle.gen_clustered(n, 1337))) / n
    assert 0.012 < r < 0.022  # analytic ~0.0163


def test_mt_matches_serial(oracle):
    data = oracle.gen_uniform(992 * 333 + 5, 11, 0.01)
    ref = oracle.compress(data)
    for t in (1, 2, 3, 8):
        assert np.array_equal(oracle.compress_mt(data, t), ref)


def test_generators_are_stable(oracle):
    """Generator spec pin (include/wah_gen.h): first words for seed 1337 never change."""
    u = oracle.gen_uniform(4, 1337, 0.5)
    v = oracle.gen_uniform(4, 1337, 0.5)
    assert np.array_equal(u, v)
    dens = np.unpackbits(oracle.gen_uniform(1 << 16, 1337, 0.01).view(np.uint8)).mean()
    assert 0.0095 < dens < 0.0105
    cl = oracle.gen_clustered(1 << 18, 1337)
    bits = np.unpackbits(cl.view(np.uint8), bitorder="little")
    flips = np.count_nonzero(bits[1:] != bits[:-1])
    mean_run = bits.size / (flips + 1)
    assert 3300 < mean_run < 5000


def test_decoder_steps_over_empty_fills(oracle):
    """A fill word of count 0 expands to nothing (decompressWords' loop runs zero times, kernels.cu:346-348); the C
    oracle and the bit-at-a-time Python statement agree on such foreign streams."""
    from tests import _oracle

    streams = [
        [0x80000000, 0xC0000000, 7, 0x80000000 | 40, 0xC0000000, 9],
        [0x80000000] * 5 + [0x7FFFFFFE, 0xC0000000 | 2] + [0xC0000000] * 3 + [1],
        [5, 0x80000000, 0x80000000, 6, 0xC0000000, 0xC0000001],
    ]
    for st in streams:
        a = np.array(st, np.uint32)
        assert np.array_equal(oracle.decompress(a), _oracle.py_decompress(a))
        assert oracle.decoded_groups(a) == sum((w & 0x3FFFFFFF) if w & 0x80000000 else 1 for w in st)
