"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bit-exact (integer work): every compressed word and every decoded word must match.
Layout follows the reference's tests: known-answer vectors of
compress() first (tests.cpp:83-239), then round-trip identities (tests.cpp:241-307),
then BASELINE.json's full-size configurations through size-independent properties.
"""
import importlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M31 = 0x7FFFFFFF
_FORCED_NO_WAIT = os.environ.get("WAH_FORCE_FALLBACK") == "1"


def _route_is(got, want):
    """The decoder a call took (DeviceDecompressor.route or wah_last_decode_route()) against the one the test expects; with
    WAH_FORCE_FALLBACK=1 in the environment every call takes the no-wait route instead (include/wah.h), and the suite is run
    that way too."""
    return got in ("no wait", 3) if _FORCED_NO_WAIT else got == want


@pytest.fixture(scope="module")
def wah():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    pkg = importlib.import_module("gpu-wah_amd")
    pkg.lib()  # raises if the HIP extension is missing: no fallback
    return pkg


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint32)


# ---------------------------------------------------------------- known-answer vectors
def test_kats_host_api(wah, kats):
    """compress() == the vectors the reference's tests store (7 must-pass + canonical answers of the 2 stale)."""
    for k in kats:
        got = wah.compress(k["data"])
        assert np.array_equal(got, k["expected"]), k["name"]
        back = wah.decompress(got)
        assert np.array_equal(back[: k["n_words"]], k["data"]), k["name"]


def test_host_calls_reuse_their_device_buffers(wah, oracle):
    """The host entry points keep their device buffers between calls (include/wah.h: wah_host_cache_release):
    growing, shrinking and released sets all give the oracle's words, and so does WAH_HOST_CACHE=0."""
    rng = np.random.default_rng(5)
    sizes = [992 * 40, 17, 992 * 300 + 5, 0, 992 * 40, 2_000_000, 31]
    cases = [(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32) & np.uint32(0x01010000) if i % 2 else
              rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)) for i, n in enumerate(sizes)]
    for rnd in range(2):
        for a in cases:
            comp = wah.compress(a)
            assert np.array_equal(comp, oracle.compress(a))
            assert np.array_equal(wah.decompress(comp)[: a.size], a)
        wah.host_cache_release()
    code = ("import importlib, numpy as np; w = importlib.import_module('gpu-wah_amd'); "
            "a = (np.arange(50000, dtype=np.uint32) * 2654435761 >> 7).astype(np.uint32) & 0x10001; "
            "[np.testing.assert_array_equal(w.decompress(w.compress(a))[:a.size], a) for _ in range(3)]; print('ok')")
    env = dict(os.environ, WAH_HOST_CACHE="0")
    out = subprocess.run(["python3", "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_kats_device_api(wah, kats):
    for k in kats:
        got = _host(wah.compress_device(_dev(k["data"])))
        assert np.array_equal(got, k["expected"]), k["name"]


def test_stale_reference_vectors_decode_to_same_bitmap(wah, kats):
    """The 93/186-word vectors of tests.cpp:66-77 are valid WAH for the same bitmap: the decoder accepts them."""
    for k in kats:
        if k["status"] == "stale":
            back = wah.decompress(k["stale_expected"])
            assert np.array_equal(back[: k["n_words"]], k["data"]), k["name"]


# ---------------------------------------------------------------- sizes and tails
@pytest.mark.parametrize("n", [0, 1, 2, 30, 31, 32, 33, 61, 62, 63, 64, 500, 991, 992, 993, 1023, 1024, 1984, 2000,
                               992 * 8, 992 * 8 + 1, 992 * 9 - 1, 31 * 1000, 261888, 262144])
def test_sizes_vs_oracle(wah, oracle, n):
    """F5 / SURVEY H1,H9: tails are zero padded to G = ceil(32n/31) groups; decoded size = ceil(31G/32)."""
    for seed, p in ((7, 0.02), (8, 0.5)):
        data = oracle.gen_uniform(n, seed, p)
        want = oracle.compress(data)
        got = wah.compress(data)
        assert np.array_equal(got, want), (n, p)
        back = wah.decompress(got)
        assert np.array_equal(back, oracle.decompress(want)), (n, p)
        assert len(back) == (n if n % 31 == 0 else n + 1)
        if n:
            gd = _host(wah.compress_device(_dev(data)))
            assert np.array_equal(gd, want), (n, p)
            bd = _host(wah.decompress_device(_dev(want), n + 1))
            assert np.array_equal(bd[:n], data), (n, p)


@pytest.mark.parametrize("n", [31, 961, 991, 993, 1983, 1984, 1985, 1984 * 3 - 1, 1984 * 3 + 5, 1984 * 8 * 3 + 993, 1984 * 8 * 7 + 17])
def test_equal_word_counts_per_lane_at_ragged_sizes(wah, oracle, n):
    """Pairs whose lanes all hold the SAME number of words (literal / zero groups in turn, p = 1/8 noise) go through the
    swizzled LDS layout of compress_pair_kernel's pass 2; here with the bitmap ending inside a lane, a segment, a pair, a tile."""
    cases = []
    for period in (2, 4):
        bits = np.zeros((32, period, 31), np.uint8)
        bits[:, 0, ::2] = 1
        w = np.packbits(bits.reshape(-1), bitorder="little").view(np.uint32)
        cases.append((f"period {period}", np.tile(w, n // w.size + 1)[:n].copy()))
    cases.append(("p 1/8", oracle.gen_uniform(n, 11, 0.125)))
    tail = oracle.gen_uniform(n, 12, 0.5)
    tail[: n // 2] = 0  # a long fill in front of incompressible words
    cases.append(("zeros then dense", tail))
    for name, data in cases:
        want = oracle.compress(data)
        got = _host(wah.compress_device(_dev(data)))
        assert got.shape == want.shape and np.array_equal(got, want), (name, n)
        back = _host(wah.decompress_device(_dev(want), n + 1))
        assert np.array_equal(back[:n], data), (name, n)


@pytest.mark.parametrize("words_behind", [0, 1, 2, 63, 64, 1023, 1024, 1151, 1152, 1153, 4095, 4096, 8191])
def test_decode_stream_ends_around_tile_boundaries(wah, oracle, words_behind):
    """The one-pass decoder's tiles are 8192 stream words (two per workgroup) and a tile's last segment reads up to 1152
    words behind it: streams of incompressible words (one group per word) that end `words_behind` words behind a tile, a
    batch of two tiles, with the last segment lacking groups -- they come out of the fill word of count 0 that the decoder
    puts behind the stream's end."""
    for tiles in (1, 2, 3):
        groups = tiles * 8192 + words_behind
        for lack in (0, 1, 17):  # bits the bitmap ends in front of a group boundary
            n = (31 * groups - lack) // 32
            if n <= 0 or (32 * n + 30) // 31 != groups:
                continue
            data = oracle.gen_uniform(n, 7 + tiles, 0.5)
            stream = oracle.compress(data)
            assert stream.size == groups  # all literals
            back = _host(wah.decompress_device(_dev(stream), n + 1))
            assert back.size in (n, n + 1) and np.array_equal(back[:n], data), (tiles, words_behind, lack)
            if back.size == n + 1:
                assert back[n] == 0
            # the same end behind a long fill: the last tile holds few words
            mixed = data.copy()
            mixed[: n // 2] = 0
            want = oracle.compress(mixed)
            back = _host(wah.decompress_device(_dev(want), n + 1))
            assert np.array_equal(back[:n], mixed), (tiles, words_behind, lack, "fill in front")


def test_decode_plain_tiles_and_lone_fills(wah, oracle):
    """decode_tile_kernel takes a short cut for PLAIN tiles -- every word of the tile's 8192 and of the 1152 behind it a literal:
    group p is word p.  A fill of count 1 (a lone all-zero or all-one group: what compress() writes for it, tests.cpp:146)
    holds one group like a literal and must not pass for one: streams of literals with single fill words of count 1 at the
    places that decide -- inside a tile, its first and last word, among the words behind a tile and behind a batch of two, in
    the tile in front of a plain one -- and foreign fills (count 0, count 2) at the same places, against the oracle's decoder."""
    rng = np.random.default_rng(11)
    tiles = 5
    base = rng.integers(1, 0x7FFFFFFF, tiles * 8192 + 700, dtype=np.int64).astype(np.uint32)  # literals, none of them 0 or 0x7FFFFFFF
    base[base == 0x7FFFFFFF] = 5
    spots = [0, 1, 4095, 4096, 8191, 8192, 8193, 8192 + 1151, 8192 + 1152, 2 * 8192 - 1, 2 * 8192, 2 * 8192 + 1151, 2 * 8192 + 1152,
             3 * 8192 + 17, 4 * 8192 - 1, 4 * 8192 + 600, base.size - 1]
    for word in (0x80000001, 0xC0000001, 0x80000002, 0xC0000000, 0x80000000):
        for spot in spots:
            stream = base.copy()
            stream[spot] = word
            want = oracle.decompress(stream)
            got = _host(wah.decompress_device(_dev(stream), want.size + 1))
            assert np.array_equal(got[: want.size], want), (hex(word), spot)
    # ... several at once, and the untouched stream (every tile plain but the last, which the stream's end makes ordinary)
    stream = base.copy()
    stream[[5, 8192 + 3, 3 * 8192 - 2, 4 * 8192 + 1]] = [0x80000001, 0xC0000001, 0xC0000001, 0x80000001]
    for st in (stream, base):
        want = oracle.decompress(st)
        assert np.array_equal(_host(wah.decompress_device(_dev(st), want.size + 1))[: want.size], want)
    # a capacity that cuts the output inside a plain tile: reported, nothing written behind it
    import torch
    want = oracle.decompress(base)
    cap = 992 * 9 + 100
    out = torch.full((cap + 64,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    dec = wah.DeviceDecompressor(base.size, cap)
    dec.out = out[:cap]
    dec.run(_dev(base))
    with pytest.raises(wah.WahError):
        dec.status()
    assert bool((out[cap:] == 0x5A5A5A5A).all())
    if dec.route == "one pass":  # (it learns the size while it writes: the part that fits is there; the other routes write nothing)
        assert np.array_equal(_host(out[:cap]), want[:cap])
    else:                        # (WAH_FORCE_FALLBACK=1: the no-wait route)
        assert _FORCED_NO_WAIT and bool((out[:cap] == 0x5A5A5A5A).all())


# ---------------------------------------------------------------- distributions
def _datasets(oracle, n):
    yield "p0.5", oracle.gen_uniform(n, 1337, 0.5)
    yield "p0.01", oracle.gen_uniform(n, 1337, 0.01)
    yield "p2^-4", oracle.gen_uniform(n, 1337, 2.0**-4)
    yield "p2^-10", oracle.gen_uniform(n, 1337, 2.0**-10)
    yield "p0.999", oracle.gen_uniform(n, 1337, 0.999)
    yield "clustered", oracle.gen_clustered(n, 1337)
    yield "zeros", np.zeros(n, np.uint32)
    yield "ones", np.full(n, 0xFFFFFFFF, np.uint32)
    rng = np.random.default_rng(5)
    mix = oracle.gen_uniform(n, 3, 0.3)
    mix[rng.random(n) < 0.6] = 0
    mix[rng.random(n) < 0.2] = 0xFFFFFFFF
    yield "mixed", mix
    # lanes with EQUAL word counts (the compress kernel's pass 2 stores those into a swizzled LDS layout):
    yield "p2^-3", oracle.gen_uniform(n, 1337, 0.125)  # four in ten segment pairs: literals but for a word or two
    half = oracle.gen_uniform(n, 1337, 0.5)  # every other segment all zeros, the others incompressible
    half[: n // 992 * 992].reshape(-1, 992)[::2] = 0
    yield "every other segment dense", half
    for period in (2, 3, 4, 32):  # one literal group, then period - 1 zero groups
        bits = np.zeros((32, period, 31), np.uint8)
        bits[:, 0, ::2] = 1
        w = np.packbits(bits.reshape(-1), bitorder="little").view(np.uint32)
        yield f"period {period}", np.tile(w, n // w.size + 1)[:n].copy()


def test_distributions_vs_oracle(wah, oracle):
    """Config-2/3/4 shaped inputs at a size the oracle finishes in seconds (4 M words = 16 MiB)."""
    n = 992 * 4228 + 77
    for name, data in _datasets(oracle, n):
        want = oracle.compress(data)
        d_in = _dev(data)
        got = _host(wah.compress_device(d_in))
        assert got.shape == want.shape and np.array_equal(got, want), name
        back = _host(wah.decompress_device(_dev(want), n + 1))
        assert np.array_equal(back[:n], data), name
        assert not back[n:].any(), name


def test_structured_blocks_vs_oracle(wah, oracle):
    """Run structures that stress wave, segment and tile boundaries (run lengths around 31/32/33/64/1024)."""
    from tests.test_oracle import _structured_block

    rng = np.random.default_rng(2024)
    data = np.concatenate([_structured_block(rng) for _ in range(96)])
    want = oracle.compress(data)
    assert np.array_equal(wah.compress(data), want)
    assert np.array_equal(_host(wah.compress_device(_dev(data))), want)
    assert np.array_equal(wah.decompress(want)[: data.size], data)


def test_generators_match_host(wah, oracle):
    """The HIP generator kernels produce exactly the bits of include/wah_gen.h on the host."""
    n = 4096 * 5 + 123
    for p in (0.5, 0.01):
        assert np.array_equal(_host(wah.gen_uniform_device(n, 1337, p)), oracle.gen_uniform(n, 1337, p))
    assert np.array_equal(_host(wah.gen_clustered_device(n, 1337)), oracle.gen_clustered(n, 1337))


# ---------------------------------------------------------------- decoder on foreign streams
def test_decoder_accepts_foreign_streams(wah, oracle):
    """The reference decoder takes any 30-bit count (kernels.cu:334): fills longer than a segment, fills that
    straddle segment boundaries, unmerged adjacent fills."""
    streams = [
        np.array([0x80000000 | 5000, 0x12345, 0xC0000000 | 3000, 0x80000001, 0x7FFFFFFE], np.uint32),
        np.array([0xC0000000 | 1, 0xC0000000 | 1, 0xC0000000 | 1022, 0x80000000 | 1023, 5], np.uint32),
        np.array([0x80000000 | (1 << 20), 1, 0xC0000000 | 70000, 2], np.uint32),
        np.array([3] * 2000 + [0x80000000 | 100] * 50 + [0xC0000000 | 7] * 300, np.uint32),
    ]
    for s in streams:
        want = oracle.decompress(s)
        got = wah.decompress(s)
        assert np.array_equal(got, want)
        gd = _host(wah.decompress_device(_dev(s), len(want)))
        assert np.array_equal(gd, want)


# ---------------------------------------------------------------- decoder: tile boundaries, odd streams
def test_decode_tile_boundaries_and_odd_streams(wah, oracle):
    """Ragged sizes around the decoder's tile (4096 words) boundaries, count-0 fills, tiles of fills only, thousands
    of output segments per tile, capacity errors."""
    rng = np.random.default_rng(77)
    # ragged sizes around tile (4096 words) and look-back group (64 tiles) boundaries, several densities
    for n in (1, 31, 992, 4096 * 3 + 5, 992 * 700 + 13, 4096 * 64 * 3 + 4097):
        for p in (0.5, 0.05, 2.0**-9):
            data = oracle.gen_uniform(n, int(rng.integers(1 << 30)), p)
            comp = oracle.compress(data)
            want = oracle.decompress(comp)
            got = _host(wah.decompress_device(_dev(comp), len(want)))
            assert np.array_equal(got, want), (n, p)
    # foreign streams: count 0, giant fills (tile totals >= 2^31), fills straddling segments, unmerged fills
    streams = [
        np.array([0x80000000 | 5000, 0x12345, 0xC0000000 | 3000, 0x80000001, 0x7FFFFFFE], np.uint32),
        np.array([0x80000000, 0xC0000000, 7, 0x80000000 | 40, 0xC0000000, 9], np.uint32),
        np.array([3] * 5000 + [0x80000000 | 100] * 4000 + [0xC0000000 | 7] * 3000 + [0x80000000] * 500 + [5], np.uint32),
        np.concatenate([np.full(9000, 0x80000000 | 3, np.uint32), np.arange(1, 9001, dtype=np.uint32)]),
    ]
    for st in streams:
        want = oracle.decompress(st)
        got = _host(wah.decompress_device(_dev(st), len(want) + 3))
        assert np.array_equal(got[: len(want)], want)
    # clustered: thousands of output segments per tile
    data = oracle.gen_clustered(992 * 3000 + 17, 99)
    comp = oracle.compress(data)
    got = _host(wah.decompress_device(_dev(comp), data.size + 1))
    assert np.array_equal(got[: data.size], data)
    # capacity below the decoded size: reported, nothing written past the capacity
    import torch
    data = oracle.gen_uniform(992 * 300, 5, 0.5)
    comp = oracle.compress(data)
    dec = wah.DeviceDecompressor(len(comp), 5000)
    guard = torch.full((5000 + 4096,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    dec.out = guard[:5000]
    dec.run(_dev(comp))
    with pytest.raises(wah.WahError):
        dec.status()
    assert bool((guard[5000:] == 0x5A5A5A5A).all())


def _random_foreign_stream(rng, n_words, max_groups):
    """Anything the reference decoder accepts: literals of any 31-bit value (0 and 0x7FFFFFFF included), fills of any
    count -- 0, 1, around the segment length, far beyond it -- in any order, unmerged."""
    kinds = rng.integers(0, 10, n_words)
    lit = rng.integers(0, 1 << 31, n_words, dtype=np.uint64).astype(np.uint32)
    lit[rng.random(n_words) < 0.1] = 0
    lit[rng.random(n_words) < 0.1] = 0x7FFFFFFF
    pool = np.array([0, 0, 1, 1, 2, 3, 30, 31, 32, 33, 63, 64, 65, 1000, 1023, 1024, 1025, 2047, 2048, 5000, 1 << 15, (1 << 16) + 1],
                    np.uint32)
    cnt = pool[rng.integers(0, len(pool), n_words)]
    bit = rng.integers(0, 2, n_words).astype(np.uint32)
    fill = np.uint32(0x80000000) | (bit << np.uint32(30)) | cnt
    st = np.where(kinds < 6, lit, fill).astype(np.uint32)
    # keep the expansion bounded: cut where the running group count would exceed max_groups
    groups = np.where(st & 0x80000000, st & 0x3FFFFFFF, 1).astype(np.int64)
    keep = int(np.searchsorted(np.cumsum(groups), max_groups, side="right"))
    return st[: max(keep, 1)]


@pytest.mark.parametrize("seed", range(6))
def test_decoder_fuzz_foreign_streams(wah, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n_words = int(rng.choice([50, 3000, 4096 * 2 + 3, 4096 * 9 + 100, 4096 * 70]))
    st = _random_foreign_stream(rng, n_words, max_groups=24_000_000)
    want = oracle.decompress(st)
    got = _host(wah.decompress_device(_dev(st), len(want) + 2))
    assert len(got) == len(want) and np.array_equal(got, want), (seed, n_words)
    # the same without empty fills: the fast routes
    st2 = st[~(((st & 0x80000000) != 0) & ((st & 0x3FFFFFFF) == 0))]
    want2 = oracle.decompress(st2)
    got2 = _host(wah.decompress_device(_dev(st2), len(want2) + 2))
    assert np.array_equal(got2, want2), (seed, n_words)
    assert np.array_equal(want2, want)  # empty fills change nothing


@pytest.mark.parametrize("seed", range(4))
def test_compress_fuzz_structured(wah, oracle, seed):
    """Random run structures: run lengths drawn around the lane, step, segment and tile sizes, literals in between."""
    rng = np.random.default_rng(2000 + seed)
    n_bits = int(rng.choice([31 * 64 * 7, 992 * 32 * 37 + 5 * 32, 992 * 32 * 15 * 9 + 992 * 32 * 3 + 64]))
    bits = np.zeros(n_bits, np.uint8)
    pos = 0
    lengths = np.array([1, 2, 30, 31, 32, 33, 61, 62, 63, 64, 65, 1983, 1984, 1985, 31 * 1024 - 1, 31 * 1024, 31 * 1024 + 1, 31 * 1024 * 3])
    while pos < n_bits:
        ln = int(lengths[rng.integers(0, len(lengths))]) + int(rng.integers(0, 3))
        kind = rng.integers(0, 3)
        if kind == 2:
            ln = min(ln, 200)
            bits[pos: pos + ln] = rng.integers(0, 2, min(ln, n_bits - pos))
        else:
            bits[pos: pos + ln] = kind
        pos += ln
    data = np.packbits(bits[: (n_bits // 32) * 32].reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
    want = oracle.compress(data)
    got = _host(wah.compress_device(_dev(data)))
    assert got.shape == want.shape and np.array_equal(got, want), seed
    back = _host(wah.decompress_device(_dev(want), data.size + 1))
    assert np.array_equal(back[: data.size], data), seed


# ---------------------------------------------------------------- stream checker
def _py_report(st):
    """Word-by-word restatement of what wah_validate_device counts."""
    pos = 0
    empty = lit = cross = unm = 0
    prev = None
    for x in (int(v) for v in st):
        fill, cnt = bool(x & 0x80000000), x & 0x3FFFFFFF
        if fill and cnt == 0:
            empty += 1
        if not fill and x in (0, 0x7FFFFFFF):
            lit += 1
        if fill and cnt and (pos % 1024) + cnt > 1024:
            cross += 1
        if fill and cnt and prev is not None and (prev & 0x80000000) and (prev & 0x3FFFFFFF) and not ((prev ^ x) & 0x40000000) and pos % 1024:
            unm += 1
        pos += cnt if fill else 1
        prev = x
    return pos, (31 * pos + 31) // 32, empty, lit, cross, unm, not (empty or lit or cross or unm)


def test_validate_streams(wah, oracle):
    # what compress() emits is segment-canonical, whatever the input
    for data in (oracle.gen_uniform(992 * 300 + 5, 3, 0.01), oracle.gen_clustered(992 * 500, 4), np.zeros(992 * 40, np.uint32),
                 np.full(5000, 0xFFFFFFFF, np.uint32), oracle.gen_uniform(4096 * 5, 5, 0.5)):
        comp = oracle.compress(data)
        r = wah.validate_device(_dev(comp))
        assert tuple(r) == _py_report(comp), r
        assert r.segment_canonical and r.groups == (32 * data.size + 30) // 31
    # foreign streams: every counter against the word-by-word restatement
    rng = np.random.default_rng(31)
    for n_words in (7, 300, 4096 * 3 + 17):
        st = _random_foreign_stream(rng, n_words, max_groups=10_000_000)
        r = wah.validate_device(_dev(st))
        assert tuple(r) == _py_report(st), (n_words, r, _py_report(st))
    hand = np.array([0x80000000 | 1000, 0x80000000 | 24, 0x80000000 | 5, 0, 0x7FFFFFFF, 0xC0000000, 0xC0000000 | 2000], np.uint32)
    r = wah.validate_device(_dev(hand))
    assert (r.empty_fills, r.fillable_literals, r.crossing_fills, r.unmerged_fills) == (1, 2, 1, 1) and not r.segment_canonical
    assert tuple(wah.validate_device(_dev(np.zeros(0, np.uint32)))) == (0, 0, 0, 0, 0, 0, True)


# ---------------------------------------------------------------- unsegmented form
def _py_merge_fills(st):
    """Word-by-word restatement of wah_merge_fills_device: drop empty fills; drop a fill whose immediate predecessor is a
    non-empty fill of the same kind inside the same block of 2^29 groups; a kept fill runs to the next kept word."""
    kept, pos = [], []
    p = 0
    prev = None
    for x in (int(v) for v in st):
        fill, cnt = bool(x & 0x80000000), x & 0x3FFFFFFF
        drop = False
        if fill and cnt == 0:
            drop = True
        elif fill and prev is not None and (prev & 0x80000000) and (prev & 0x3FFFFFFF) and not ((prev ^ x) & 0x40000000):
            drop = ((p - (prev & 0x3FFFFFFF)) >> 29) == ((p + cnt - 1) >> 29)
        if not drop:
            kept.append(x)
            pos.append(p)
        p += cnt if fill else 1
        prev = x
    pos.append(p)
    out = [(x & 0xC0000000) | (pos[i + 1] - pos[i]) if (x & 0x80000000) and (x & 0x3FFFFFFF) else x for i, x in enumerate(kept)]
    return np.array(out, np.uint32)


def test_merge_fills_unsegmented_form(wah, oracle):
    rng = np.random.default_rng(8)
    cases = [oracle.compress(np.zeros(992 * 700 + 3, np.uint32)), oracle.compress(oracle.gen_uniform(992 * 900, 2, 2.0**-14)),
             oracle.compress(oracle.gen_clustered(992 * 600 + 11, 3, 50000)), oracle.compress(oracle.gen_uniform(992 * 40, 4, 0.3)),
             _random_foreign_stream(rng, 4096 * 3 + 9, 30_000_000), np.zeros(0, np.uint32),
             np.array([0x80000000 | ((1 << 29) - 4), 0x80000000 | 8, 0x80000000 | 9, 0xC0000000 | 1, 0xC0000000, 0xC0000000 | 2], np.uint32)]
    for st in cases:
        got = _host(wah.merge_fills_device(_dev(st)))
        assert np.array_equal(got, _py_merge_fills(st))
        if len(st) and oracle.decoded_groups(st) < 40_000_000:
            assert np.array_equal(oracle.decompress(got), oracle.decompress(st))
    # an all-zero bitmap: one word per segment before, one word after
    z = oracle.compress(np.zeros(992 * 700, np.uint32))
    assert len(z) == 700 and np.array_equal(_host(wah.merge_fills_device(_dev(z))), [0x80000000 | 700 * 1024])
    # the merged stream is no longer what the reference's encoder emits, and the checker says so
    assert not wah.validate_device(wah.merge_fills_device(_dev(z))).segment_canonical


def test_unsegmented_encoder_mode(wah, oracle):
    """wah_compress_device_ex(..., WAH_UNSEGMENTED) == merge_fills(compress(x)), bit for bit, in the one pass: runs that
    cross segments, waves, tiles, rows of tiles, and (last case) a superrow of tiles and the 2^29-group cut."""
    import torch

    rng = np.random.default_rng(11)

    def islands(n, k, seed):  # zeros (or ones) with k short random islands: long runs through many transparent tiles
        x = np.zeros(n, np.uint32) if seed % 2 else np.full(n, 0xFFFFFFFF, np.uint32)
        for p in np.random.default_rng(seed).integers(0, n - 40, k):
            x[p: p + 37] = np.random.default_rng(seed + int(p)).integers(0, 2**32, 37, dtype=np.uint64).astype(np.uint32)
        return x

    alternating = np.zeros(992 * 64, np.uint32)
    alternating.reshape(64, 992)[1::2] = 0xFFFFFFFF
    cases = [np.zeros(992 * 700 + 3, np.uint32), np.full(992 * 333, 0xFFFFFFFF, np.uint32), oracle.gen_uniform(992 * 900, 2, 2.0**-14),
             oracle.gen_clustered(992 * 600 + 11, 3, 50000), oracle.gen_uniform(992 * 40, 4, 0.3), alternating,
             islands(992 * 30000, 9, 1), islands(992 * 30000 + 77, 40, 2), islands(992 * 9000, 3, 3), np.zeros(31, np.uint32),
             oracle.gen_clustered(992 * 30000, 5, 2_000_000)]
    for k in (2, 3, 24, 48, 50):  # runs of k whole segments, zeros and ones in turn: run ends ON pair, wave and tile boundaries
        blocks = np.zeros((130, k, 992), np.uint32)
        blocks[1::2] = 0xFFFFFFFF
        cases.append(blocks.reshape(-1)[: 992 * (130 * k) - (5 if k == 3 else 0)].copy())
    for x in cases:
        comp = wah.DeviceCompressor(len(x), unsegmented=True)
        comp.run(_dev(x))
        got = _host(comp.result())
        want = _py_merge_fills(oracle.compress(x))
        assert np.array_equal(got, want), (len(x), len(got), len(want))
        back = _host(wah.decompress_device(comp.result().clone(), len(x) + 1))
        assert np.array_equal(back[: len(x)], x)
        del comp
    # 600 000 all-zero segments: the run crosses rows and a superrow of tiles and is cut at 2^29 groups (segment 524 288)
    n = 992 * 600000
    z = torch.zeros(n, dtype=torch.int32, device="cuda")
    comp = wah.DeviceCompressor(n, unsegmented=True)
    comp.run(z)
    assert _host(comp.result()).tolist() == [0x80000000 | (1 << 29), 0x80000000 | ((600000 - 524288) * 1024)]
    z[992 * 524288 - 5] = 7  # a literal just in front of the cut, ones behind it
    z[992 * 524288:] = -1
    comp.run(z)
    got = _host(comp.result())
    want = _py_merge_fills(oracle.compress(z.cpu().numpy().view(np.uint32)))
    assert np.array_equal(got, want)
    # an unsegmented stream has no segment index, and unknown flags are refused
    with pytest.raises(wah.WahError):
        wah.DeviceCompressor(992, indexed=True, unsegmented=True)
    assert wah.lib().wah_compress_device_ex(z.data_ptr(), 992, comp.out.data_ptr(), comp.capacity, comp.count.data_ptr(), 6,
                                            comp.workspace.data_ptr(), comp.ws_bytes, None) == -1


# ---------------------------------------------------------------- host decompress: kept output buffer; count slots, streams
def test_host_decompress_kept_output_buffer(wah, oracle):
    """decompress() expands right behind its scan into the output buffer kept from the call before, and falls back to
    scan -> allocate -> expand when that buffer is too small: small, large, small again, then an empty stream."""
    for n, p in ((992 * 3, 0.3), (992 * 4000 + 5, 0.01), (31, 0.5), (992 * 900, 2.0**-9), (992 * 4000 + 5, 0.5), (1, 0.5)):
        data = oracle.gen_uniform(n, n % 97, p)
        comp = oracle.compress(data)
        back = wah.decompress(comp)
        assert np.array_equal(back, oracle.decompress(comp)), (n, p)
    assert len(wah.decompress(np.zeros(0, np.uint32))) == 0
    # a malformed stream (groups beyond 2^47) is still refused on the short cut
    with pytest.raises(wah.WahError):
        wah.decompress(np.full(1 << 18, 0xBFFFFFFF, np.uint32))


def test_host_decompress_first_call_is_one_pass(wah, oracle):
    """A process's first decompress() -- no output buffer kept -- sizes one from a sample of the stream (include/wah.h) and
    decodes in the single pass it would take with a kept buffer (the reference's order, scan -> read the size back -> allocate
    -> expand, decompress.cu:72-100, only when the prediction was too small).  The library says which decoder its last launch
    was: the one-pass decoder after a first call on an incompressible stream; after a first call whose sample MISSES a giant
    fill (a foreign stream: every sampled word is a literal) the expand-only launch of the by-the-book path -- and the same
    words either way."""
    lib = wah.lib()
    n = 992 * 3000 + 17
    data = oracle.gen_uniform(n, 5, 0.5)
    comp = oracle.compress(data)
    want = oracle.decompress(comp)
    wah.host_cache_release()
    assert np.array_equal(wah.decompress(comp), want)
    assert _route_is(lib.wah_last_decode_route(), 1), "first call: one pass into a buffer sized by the sample"
    # 2 * 65536 + 2 words: the sample takes every second one; the giant fill sits on an odd index
    foreign = np.full(2 * 65536 + 2, 0x2AAAAAAA, np.uint32)
    foreign[1001] = 0x80000000 | 40_000_000  # a zero fill of 4e7 groups
    want = oracle.decompress(foreign)
    wah.host_cache_release()
    got = wah.decompress(foreign)
    assert np.array_equal(got, want)
    assert lib.wah_last_decode_route() == 2, "prediction too small: scan, allocate, expand (the expand-only launch)"
    assert np.array_equal(wah.decompress(foreign), want)  # (now into the kept buffer)
    assert _route_is(lib.wah_last_decode_route(), 1)
    # ... and a stream of 8 to 128 groups per word (here: one bit in 2^10, about 17) goes by the scan + expansion launches, which
    # are the faster decoder for it -- decided from the same sample
    mid = oracle.gen_uniform(992 * 3000, 6, 2.0 ** -10)
    cm = oracle.compress(mid)
    assert np.array_equal(wah.decompress(cm)[: mid.size], mid)
    assert _route_is(lib.wah_last_decode_route(), 2)
    wah.host_cache_release()


def test_compress_count_slots_and_streams(wah, oracle):
    """DeviceCompressor.run(count=slot, stream=s): launches of several compressors on several streams, every launch
    writing its C into its own slot (the columns workload of bench.py)."""
    import torch

    n = 992 * 257 + 3
    cols = [oracle.gen_uniform(n, 100 + i, (0.5, 0.01, 2.0**-8)[i % 3]) for i in range(6)]
    d_cols = [_dev(c) for c in cols]
    comps = [wah.DeviceCompressor(n) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    slots = torch.zeros(len(cols), dtype=torch.int64, device="cuda")
    main = torch.cuda.current_stream()
    results = []
    for i, d in enumerate(d_cols):
        c, st = comps[i % 2], streams[i % 2]
        st.wait_stream(main)
        c.run(d, stream=st, count=slots[i:i + 1])
        with torch.cuda.stream(st):
            results.append(c.out[: c.capacity].clone())  # (the compressor's buffer is reused by its next column)
    for st in streams:
        main.wait_stream(st)
    torch.cuda.synchronize()
    for c in comps:
        c.status()
    for i, col in enumerate(cols):
        want = oracle.compress(col)
        assert int(slots[i].item()) == len(want)
        assert np.array_equal(_host(results[i][: len(want)]), want), i
    with pytest.raises(wah.WahError):
        comps[0].run(d_cols[0], count=torch.zeros(2, dtype=torch.int64, device="cuda"))


# ---------------------------------------------------------------- the route on which nobody waits for anybody
def test_no_wait_route(wah, oracle, monkeypatch):
    """WAH_NO_WAIT / WAH_FORCE_FALLBACK=1: count, scan, place -- three launches, no in-kernel wait -- give the stream of the
    one-launch kernel bit for bit: tails, several tiles, a table longer than one scan round, the index, the host entry
    point, and the same workspace serving both routes in turn."""
    import torch

    sizes = (1, 31, 992, 992 * 16 + 5, 992 * 700 + 13, 992 * 40000 + 1)  # 40 000 segments: 2 500 tiles, three scan rounds
    for n in sizes:
        for seed, p in ((3, 0.01), (4, 0.5)):
            data = oracle.gen_uniform(n, seed, p)
            want = oracle.compress(data)
            comp = wah.DeviceCompressor(n, no_wait=True)
            comp.run(_dev(data))
            assert np.array_equal(_host(comp.result()), want), (n, p)
            del comp
    data = oracle.gen_clustered(992 * 3000 + 17, 99)
    want = oracle.compress(data)
    d = _dev(data)
    # one workspace, both routes in turn (the table lives in the unsegmented mode's scan area and must not disturb it)
    normal = wah.DeviceCompressor(len(data))
    normal.run(d)
    assert np.array_equal(_host(normal.result()), want)
    lib = wah.lib()
    for flags in (2, 0, 1, 2, 0):
        rc = lib.wah_compress_device_ex(d.data_ptr(), len(data), normal.out.data_ptr(), normal.capacity, normal.count.data_ptr(), flags,
                                        normal.workspace.data_ptr(), normal.ws_bytes, None)
        assert rc == 0
        got = _host(normal.result())
        assert np.array_equal(got, want if flags != 1 else _py_merge_fills(want)), flags
    # the unsegmented mode has its no-wait route too (count incl. (T, L) -> scan -> place), an unknown flag is refused
    assert lib.wah_compress_device_ex(d.data_ptr(), len(data), normal.out.data_ptr(), normal.capacity, normal.count.data_ptr(), 3,
                                      normal.workspace.data_ptr(), normal.ws_bytes, None) == 0
    assert np.array_equal(_host(normal.result()), _py_merge_fills(want))
    assert lib.wah_compress_device_ex(d.data_ptr(), len(data), normal.out.data_ptr(), normal.capacity, normal.count.data_ptr(), 4,
                                      normal.workspace.data_ptr(), normal.ws_bytes, None) == -1
    # an output capacity below C is reported, nothing is written past it
    guard = torch.full((len(data),), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    assert lib.wah_compress_device_ex(d.data_ptr(), len(data), guard.data_ptr(), 100, normal.count.data_ptr(), 2,
                                      normal.workspace.data_ptr(), normal.ws_bytes, None) == 0
    assert lib.wah_compress_status(normal.workspace.data_ptr(), None) == -4
    assert bool((guard[100:] == 0x5A5A5A5A).all())
    # the environment switch sends every plain compress launch this way: device API with the index, host entry point
    monkeypatch.setenv("WAH_FORCE_FALLBACK", "1")
    indexed = wah.DeviceCompressor(len(data), indexed=True)
    indexed.run(d)
    assert np.array_equal(_host(indexed.result()), want)
    offs = indexed.seg_offsets.cpu().numpy()
    assert offs[0] == 0 and offs[-1] == len(want) and bool(np.all(np.diff(offs) >= 1))
    assert np.array_equal(wah.compress(data), want)
    monkeypatch.delenv("WAH_FORCE_FALLBACK")
    normal.run(d)
    assert np.array_equal(_host(normal.result()), want)
    # the decoder's sums pass: per-tile totals + one scan launch instead of the one-launch row scan; one workspace, both routes
    stream = _dev(want)
    ref = oracle.decompress(want)
    dec = wah.DeviceDecompressor(len(want), len(data) + 1)
    for flags in (2, 0, 2, 0):
        rc = lib.wah_decompress_device_ex(stream.data_ptr(), len(want), dec.out.data_ptr(), dec.capacity, dec.info.data_ptr(), flags,
                                          dec.workspace.data_ptr(), dec.ws_bytes, None)
        assert rc == 0
        assert np.array_equal(_host(dec.result()), ref), flags
    assert lib.wah_decompress_device_ex(stream.data_ptr(), len(want), dec.out.data_ptr(), dec.capacity, dec.info.data_ptr(), 1,
                                        dec.workspace.data_ptr(), dec.ws_bytes, None) == -1
    # a stream whose totals saturate is still reported
    huge = _dev(np.full(5000, 0xBFFFFFFF, dtype=np.uint32))
    big = wah.DeviceDecompressor(5000, 1024, no_wait=True)
    big.run(huge)
    with pytest.raises(wah.WahError, match="capacity|stream"):
        big.status()


def test_timeout_takes_the_no_wait_route(wah, oracle, monkeypatch):
    """compress() / decompress() after a WAH_ERR_TIMEOUT of their one-launch kernels (forced by the test hook: the first launch
    is treated as if a bounded wait had expired): the no-wait route delivers the same result, the device-phase timing covers
    it, and the kept workspace is usable by the next ordinary call."""
    data = oracle.gen_uniform(992 * 300 + 7, 21, 0.02)
    want = oracle.compress(data)
    monkeypatch.setenv("WAH_FAULT_INJECT", "timeout")
    got, t = wah.compress(data, with_timings=True)
    assert np.array_equal(got, want) and t.device_ms > 0
    back, t = wah.decompress(got, with_timings=True)
    assert np.array_equal(back[: len(data)], data) and t.device_ms > 0
    monkeypatch.delenv("WAH_FAULT_INJECT")
    assert np.array_equal(wah.compress(data), want)
    assert np.array_equal(wah.decompress(want)[: len(data)], data)


# ---------------------------------------------------------------- bitwise operations on compressed bitmaps
def test_bitops_on_compressed_bitmaps(wah, oracle):
    """wah_bitop_device(op, A, B) == compress(decompress(A) op decompress(B)), for whole-segment and ragged lengths."""
    ops = {"and": np.bitwise_and, "or": np.bitwise_or, "xor": np.bitwise_xor, "andnot": lambda x, y: x & ~y}
    for n in (992 * 64, 992 * 300 + 17, 5):
        a = oracle.gen_uniform(n, 11, 0.05)
        b = oracle.gen_clustered(n, 12, 700)
        ca, cb = oracle.compress(a), oracle.compress(b)
        for name, fn in ops.items():
            want = oracle.compress(fn(a, b).astype(np.uint32))
            got = _host(wah.bitop_device(name, _dev(ca), _dev(cb), n))
            assert got.shape == want.shape and np.array_equal(got, want), (n, name)
    # operands that do not describe bitmaps of the stated length are refused
    a = oracle.gen_uniform(992 * 10, 1, 0.1)
    with pytest.raises(wah.WahError):
        wah.bitop_device("and", _dev(oracle.compress(a)), _dev(oracle.compress(a[:992 * 9])), a.size)


def test_bitops_through_the_segment_indexes(wah, oracle):
    """wah_bitop_indexed_device(op, A + index, B + index) == compress(A op B), and the result's own index is right
    (so results chain: (A and B) or C)."""
    ops = {"and": np.bitwise_and, "or": np.bitwise_or, "xor": np.bitwise_xor, "andnot": lambda x, y: x & ~y}
    for n in (992 * 64, 992 * 300 + 17, 5, 992):
        a = oracle.gen_uniform(n, 21, 0.05)
        b = oracle.gen_clustered(n, 22, 700)
        c = oracle.gen_uniform(n, 23, 0.5)
        (sa, oa), (sb, ob), (sc, oc) = (_indexed_stream(wah, _dev(x)) for x in (a, b, c))
        for name, fn in ops.items():
            want = oracle.compress(fn(a, b).astype(np.uint32))
            got, offs = wah.bitop_indexed_device(name, sa, oa, sb, ob, n)
            assert got.numel() == want.size and np.array_equal(_host(got), want), (n, name)
            ref_stream, ref_offs = _indexed_stream(wah, _dev(fn(a, b).astype(np.uint32)))
            assert np.array_equal(offs.cpu().numpy(), ref_offs.cpu().numpy()), (n, name)
        ab, oab = wah.bitop_indexed_device("and", sa, oa, sb, ob, n)
        abc, _ = wah.bitop_indexed_device("or", ab.clone(), oab.clone(), sc, oc, n)
        assert np.array_equal(_host(abc), oracle.compress(((a & b) | c).astype(np.uint32))), n
    # operands of another length, or an index that does not belong to the stream, are refused
    a = oracle.gen_uniform(992 * 10, 1, 0.1)
    sa, oa = _indexed_stream(wah, _dev(a))
    sb, ob = _indexed_stream(wah, _dev(a[: 992 * 9]))
    with pytest.raises(wah.WahError):
        wah.bitop_indexed_device("and", sa, oa, sb, torch_pad(ob, oa.numel()), a.size)
    bad = oa.clone()
    bad[3] += 1
    with pytest.raises(wah.WahError):
        wah.bitop_indexed_device("xor", sa, bad, sa, oa, a.size)
    with pytest.raises(wah.WahError):
        wah.bitop_indexed_device("xor", sa, oa, sa, bad, a.size)


def test_many_operand_bitops_through_the_segment_indexes(wah, oracle):
    """wah_bitop_many_indexed_device: A op B op C ... in one combining pass == numpy on the bitmaps, for 1..8 operands."""
    fold = {"and": lambda xs: np.bitwise_and.reduce(xs), "or": lambda xs: np.bitwise_or.reduce(xs),
            "xor": lambda xs: np.bitwise_xor.reduce(xs), "andnot": lambda xs: xs[0] & ~np.bitwise_or.reduce(xs[1:]) if len(xs) > 1 else xs[0]}
    for n in (992 * 40 + 9, 7, 992 * 257):
        maps = [oracle.gen_uniform(n, 30 + j, 0.3) if j % 3 == 0 else oracle.gen_clustered(n, 30 + j, 200 + 150 * j) if j % 3 == 1
                else oracle.gen_uniform(n, 30 + j, 0.9) for j in range(8)]
        ops = [_indexed_stream(wah, _dev(m)) for m in maps]
        for k in (1, 2, 3, 5, 8):
            for name, fn in fold.items():
                want = oracle.compress(fn(np.stack(maps[:k])).astype(np.uint32))
                got, offs = wah.bitop_many_indexed_device(name, ops[:k], n)
                assert got.numel() == want.size and np.array_equal(_host(got), want), (n, k, name)
    with pytest.raises(wah.WahError):
        wah.bitop_many_indexed_device("and", ops * 2, n)      # 16 operands
    bad = ops[2][1].clone()
    bad[1] += 1
    with pytest.raises(wah.WahError):
        wah.bitop_many_indexed_device("or", [ops[0], ops[1], (ops[2][0], bad)], n)


def test_bitops_by_run_merge(wah, oracle):
    """Operands of few words per segment (include/wah.h: WAH_BITOP_ROUTE_RUNS): the operands' runs merged in the compressed
    domain, one lane per segment -- the same words and the same index as compress() of the combined bitmap, for every operation,
    1..8 operands, ragged lengths (a last segment of few groups, a last group of few bits), a dense stretch inside sparse
    operands (a tile whose words do not fit the LDS is read from global memory), all-zero and all-one bitmaps; corrupt operands
    and too small an output are refused."""
    import torch

    lib = wah.lib()
    fold = {"and": lambda xs: np.bitwise_and.reduce(xs), "or": lambda xs: np.bitwise_or.reduce(xs),
            "xor": lambda xs: np.bitwise_xor.reduce(xs), "andnot": lambda xs: xs[0] & ~np.bitwise_or.reduce(xs[1:]) if len(xs) > 1 else xs[0]}
    for n in (992 * 700 + 9, 992 * 256, 992 * 64 + 991, 33, 1):
        maps = [oracle.gen_clustered(n, 50 + j, 6000 + 1500 * j) if j % 2 == 0 else oracle.gen_uniform(n, 50 + j, 2.0 ** -(13 + j % 3)) for j in range(8)]
        maps[3] = np.zeros(n, np.uint32)
        maps[5] = np.full(n, 0xFFFFFFFF, np.uint32)
        if n > 992 * 300:
            maps[0] = maps[0].copy()
            maps[0][992 * 300: 992 * 308] = oracle.gen_uniform(992 * 8, 3, 0.5)  # 8 incompressible segments: a tile that is not staged
        ops = [_indexed_stream(wah, _dev(m)) for m in maps]
        for k in (1, 2, 3, 4, 8):
            for name, fn in fold.items():
                combined = fn(np.stack(maps[:k])).astype(np.uint32)
                want = oracle.compress(combined)
                got, offs = wah.bitop_many_indexed_device(name, ops[:k], n)
                assert lib.wah_last_bitop_route() == 1, (n, k, "run merge expected")
                assert got.numel() == want.size and np.array_equal(_host(got), want), (n, k, name)
                _, ref_offs = _indexed_stream(wah, _dev(combined))
                assert np.array_equal(offs.cpu().numpy(), ref_offs.cpu().numpy()), (n, k, name)
        for name, fn in fold.items():
            want = oracle.compress(fn(np.stack(maps[:2])).astype(np.uint32))
            got, offs = wah.bitop_indexed_device(name, *ops[0], *ops[1], n)
            assert lib.wah_last_bitop_route() == 1 and np.array_equal(_host(got), want), (n, name)
        # results chain (their index is right): (A and B) or C
        ab, oab = wah.bitop_indexed_device("and", *ops[0], *ops[1], n)
        abc, _ = wah.bitop_indexed_device("or", ab.clone(), oab.clone(), *ops[2], n)
        assert np.array_equal(_host(abc), oracle.compress(((maps[0] & maps[1]) | maps[2]).astype(np.uint32))), n
    # a dense operand beside it: the other route, the same call
    n = 992 * 64
    a, b = oracle.gen_clustered(n, 1, 4000), oracle.gen_uniform(n, 2, 0.3)
    (sa, oa), (sb, ob) = (_indexed_stream(wah, _dev(x)) for x in (a, b))
    got, _ = wah.bitop_indexed_device("xor", sa, oa, sb, ob, n)
    assert lib.wah_last_bitop_route() == 2 and np.array_equal(_host(got), oracle.compress(a ^ b))
    # refused: an index that is not the stream's, a fill that runs past its segment, an empty fill, too small an output
    (sc_, oc) = _indexed_stream(wah, _dev(oracle.gen_clustered(n, 3, 4000)))
    bad = oa.clone()
    bad[3] += 1
    for args in ((sa, bad, sc_, oc), (sc_, oc, sa, bad)):
        with pytest.raises(wah.WahError):
            wah.bitop_indexed_device("and", *args, n)
    zeros = np.zeros(n, np.uint32)
    sz, oz = _indexed_stream(wah, _dev(zeros))                     # 64 words: one zero fill of 1024 groups per segment
    for word in (0x80000000 | 1025, 0x80000000, 0x80000000 | 1023):
        broken = sz.clone()
        broken[5] = word - (1 << 32) if word >= (1 << 31) else word
        with pytest.raises(wah.WahError):
            wah.bitop_indexed_device("or", broken, oz, sc_, oc, n)
    small = torch.empty(3, dtype=torch.int32, device="cuda")
    with pytest.raises(wah.WahError):
        wah.bitop_indexed_device("or", sa, oa, sc_, oc, n, out=small)


def _run_structured_bitmap(rng, n_words, mean_run_groups):
    """A bitmap of long runs with a literal at most of their ends: run lengths (in bits) around the group, step and segment sizes."""
    n_bits = n_words * 32
    bits = np.zeros(n_bits, np.uint8)
    pos = 0
    lengths = np.array([1, 30, 31, 32, 61, 62, 63, 31 * 64, 31 * 64 + 1, 31 * 1023, 31 * 1024, 31 * 1024 + 1, 31 * 2048 + 5])
    while pos < n_bits:
        if rng.random() < 0.5:
            ln = int(rng.geometric(1.0 / (31 * mean_run_groups)))
        else:
            ln = int(lengths[rng.integers(0, len(lengths))])
        kind = int(rng.integers(0, 8))
        if kind >= 6:  # a few random bits
            ln = min(ln, 70)
            bits[pos: pos + ln] = rng.integers(0, 2, min(ln, n_bits - pos))
        else:
            bits[pos: pos + ln] = kind & 1
        pos += ln
    return np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()


@pytest.mark.parametrize("seed", range(6))
def test_bitops_by_run_merge_fuzz(wah, oracle, seed):
    """Random run structures through the run merge: 2 .. 5 operands, every operation, bitmaps of a few segments up to a few
    tiles of 256 segments, with a ragged end -- against compress() of the combined bitmaps, words and index."""
    rng = np.random.default_rng(4000 + seed)
    lib = wah.lib()
    fold = {"and": lambda xs: np.bitwise_and.reduce(xs), "or": lambda xs: np.bitwise_or.reduce(xs),
            "xor": lambda xs: np.bitwise_xor.reduce(xs), "andnot": lambda xs: xs[0] & ~np.bitwise_or.reduce(xs[1:])}
    n = int(rng.choice([992 * 3 + 1, 992 * 40, 992 * 257 + 500, 992 * 600 + 31]))
    k = int(rng.integers(2, 6))
    maps = [_run_structured_bitmap(rng, n, int(rng.choice([60, 200, 900]))) for _ in range(k)]
    ops = [_indexed_stream(wah, _dev(m)) for m in maps]
    for name, fn in fold.items():
        combined = fn(np.stack(maps)).astype(np.uint32)
        want = oracle.compress(combined)
        got, offs = wah.bitop_many_indexed_device(name, ops, n)
        route = lib.wah_last_bitop_route()
        assert got.numel() == want.size and np.array_equal(_host(got), want), (seed, n, k, name, route)
        _, ref_offs = _indexed_stream(wah, _dev(combined))
        assert np.array_equal(offs.cpu().numpy(), ref_offs.cpu().numpy()), (seed, n, k, name, route)
    assert route == 1, (seed, n, k, "these operands are meant for the run merge", sum(int(o[0].numel()) for o in ops), n // 992)


def torch_pad(t, n):
    """t extended to n entries by repeating its last one (an index that claims more segments than the stream has)."""
    import torch

    return torch.cat([t, t[-1:].expand(n - t.numel())]) if t.numel() < n else t


# ---------------------------------------------------------------- API behaviour
def test_reusable_workspace_and_indexed_output(wah, oracle):
    """Same DeviceCompressor run repeatedly (nothing is cleared between launches: launch epochs) + the segment index."""
    import torch

    n = 992 * 1000
    comp = wah.DeviceCompressor(n, indexed=True)
    for seed in (1, 2, 3):
        data = oracle.gen_uniform(n, seed, 0.01)
        comp.run(_dev(data))
        got = _host(comp.result())
        want = oracle.compress(data)
        assert np.array_equal(got, want)
        offs = comp.seg_offsets.cpu().numpy()
        assert offs[0] == 0 and offs[-1] == len(want) and np.all(np.diff(offs) >= 1)
        # segment s alone compresses to exactly out[offs[s]:offs[s+1]] (F4: segments are independent)
        for s in (0, 17, 999):
            assert np.array_equal(oracle.compress(data[992 * s: 992 * (s + 1)]), want[offs[s]: offs[s + 1]])
    torch.cuda.synchronize()


def test_workspace_serves_different_sizes_in_turn(wah, oracle):
    """One workspace, bitmaps of very different sizes one after the other (1, 2 and 5 segments per wavefront, so the
    tile <-> granule mapping changes between launches): nothing is cleared in between, launch epochs keep the
    launches apart (include/wah.h: wah_workspace_init_device)."""
    big = 992 * 30000
    comp = wah.DeviceCompressor(big)
    dec = wah.DeviceDecompressor(wah.max_compressed_words(big), big + 1)
    for i, n in enumerate((big, 992 * 7, 992 * 3000 + 5, 31, big, 992 * 5000, 1)):
        data = oracle.gen_uniform(n, 100 + i, (0.01, 0.5, 0.0001)[i % 3])
        comp.run(_dev(data), n_words=n)
        got = comp.result()
        assert np.array_equal(_host(got), oracle.compress(data)), n
        dec.run(got.clone(), c_words=got.numel())
        assert np.array_equal(_host(dec.result())[:n], data), n


def test_decode_long_fills_inside_dense_data(wah, oracle):
    """Incompressible data with long holes: the stream is short enough against its output for the one-pass decoder, whose
    tiles with a giant fill (thousands of output segments out of one word) go to the second launch in parts."""
    n = 992 * 9000
    x = oracle.gen_uniform(n, 77, 0.5)
    x[992 * 100 + 5: 992 * 2100 - 7] = 0                      # 2000 segments of zeros
    x[992 * 4000: 992 * 4000 + 40] = 0xFFFFFFFF                # a short run of ones
    x[992 * 6000 + 1: 992 * 7500] = 0xFFFFFFFF                 # 1500 segments of ones
    x[992 * 8000: 992 * 8300] = oracle.gen_clustered(992 * 300, 78)  # a clustered stretch: tiles that expand 60 x
    st = oracle.compress(x)
    assert (n + 1) // 8 <= len(st)                              # (the one-pass route's condition)
    back = _host(wah.decompress_device(_dev(st), n + 1))
    assert np.array_equal(back[:n], x)
    # ... and the classic form of the same bitmap (fills that cross segments: two giant words)
    un = _py_merge_fills(st)
    back = _host(wah.decompress_device(_dev(un), n + 1))
    assert np.array_equal(back[:n], x)


def test_decode_mostly_empty_bitmap_with_dense_islands(wah, oracle):
    """Classic (unsegmented) WAH of a mostly empty bitmap: a short stream -- the two-launch route -- whose tiles around the
    islands hold fills of millions of groups.  Those tiles are put on a list by the sums pass and shared out over the
    workgroups of a launch of their own (one workgroup used to expand such a tile alone: 3.6 ms for this shape at 1 GiB,
    0.25 ms now); here the parity of that path, scan and no-wait routes."""
    import torch

    n = 992 * 40000
    x = np.zeros(n, np.uint32)
    x[992 * 9000: 992 * 9300] = oracle.gen_uniform(992 * 300, 5, 0.5)
    x[992 * 30000 + 7: 992 * 30200] = oracle.gen_uniform(992 * 200 - 7, 6, 0.5)
    x[992 * 39000:] = 0xFFFFFFFF
    st = _py_merge_fills(oracle.compress(x))
    assert len(st) * 8 < n  # (highly compressed: not the one-pass decoder's case)
    for kw, route in (({}, "one pass"), ({"two_launches": True}, "two launches"), ({"no_wait": True}, "no wait")):
        dec = wah.DeviceDecompressor(len(st), n + 1, **kw)
        dec.run(_dev(st))
        assert _route_is(dec.route, route)
        assert np.array_equal(_host(dec.result())[:n], x), route
        dec.run(_dev(st))  # (the list's counters change hands from launch to launch)
        assert np.array_equal(_host(dec.result())[:n], x), route


def test_decoder_route_is_a_property_of_the_stream(wah, oracle):
    """The one-pass decoder is correct for EVERY stream: decode_tile_kernel decides tile by tile whether it expands a tile
    itself or puts it on the list the launch behind it shares out (include/wah.h).  An incompressible stream with 64 times the
    capacity it needs, a highly compressed one with exactly what it needs, a mix of both: the default takes the one pass
    (wah_last_decode_route) and the words are the oracle's.  Where the capacity allows 7 to 40 words of output per word of
    stream the default is the two launches (faster for streams of that kind), and WAH_ONE_PASS / WAH_TWO_LAUNCHES overrule
    it either way with the same words.  The other routes by necessity: WAH_NO_WAIT, a stream that is only 4-byte aligned.
    And what a capacity that is too small leaves behind, route by route (d_out is undefined then, include/wah.h: pinned here
    so that a change is seen)."""
    import torch

    n = 992 * 2600
    dense = oracle.gen_uniform(n, 3, 0.5)
    clustered = oracle.gen_clustered(n, 4)
    mixed = dense.copy()
    mixed[992 * 700: 992 * 1900] = clustered[992 * 700: 992 * 1900]
    middling = oracle.gen_uniform(n, 5, 2.0 ** -9)  # about 9 groups per word
    st = oracle.compress(middling)
    want = oracle.decompress(st)
    for kw, route in (({}, "two launches"), ({"one_pass": True}, "one pass"), ({"two_launches": True}, "two launches")):
        dec = wah.DeviceDecompressor(len(st), n + 1, **kw)
        dec.run(_dev(st))
        assert _route_is(dec.route, route), (kw, dec.route)
        assert np.array_equal(_host(dec.result()), want), kw
    dec = wah.DeviceDecompressor(len(st), 64 * n)  # ... and with a capacity that says nothing, the decoder that is right for anything
    dec.run(_dev(st))
    assert _route_is(dec.route, "one pass") and np.array_equal(_host(dec.result()), want)
    with pytest.raises(wah.WahError):
        wah.DeviceDecompressor(len(st), n + 1, one_pass=True, two_launches=True).run(_dev(st))
    for name, x, cap in (("dense, 64 x the capacity", dense, 64 * n), ("clustered", clustered, n + 1), ("mixed", mixed, 3 * n)):
        st = oracle.compress(x)
        want = oracle.decompress(st)
        for kw, route in (({}, "one pass"), ({"two_launches": True}, "two launches"), ({"no_wait": True}, "no wait")):
            dec = wah.DeviceDecompressor(len(st), cap, **kw)
            dec.run(_dev(st))
            assert _route_is(dec.route, route), (name, dec.route)
            assert np.array_equal(_host(dec.result()), want), (name, route)
            dec.run(_dev(st))
            assert np.array_equal(_host(dec.result()), want), (name, route, "again")
        # 4-byte aligned only: the scan + expansion launches, whatever was asked for
        shifted = torch.empty(len(st) + 1, dtype=torch.int32, device="cuda")[1:]
        shifted.copy_(_dev(st))
        dec = wah.DeviceDecompressor(len(st), cap)
        dec.run(shifted)
        assert _route_is(dec.route, "two launches") and np.array_equal(_host(dec.result()), want), name
        # too small a capacity: reported by both, nothing written behind it; the two launches write nothing at all, the one-pass
        # decoder what fits of the tiles it expands itself (all of them in the incompressible stream; the tiles on its list are
        # expanded by the launch behind it, which knows the size and writes nothing)
        short = 992 * 1000 + 17
        for kw in ({}, {"two_launches": True}):
            dec = wah.DeviceDecompressor(len(st), short, **kw)
            whole = torch.full((short + 64,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            dec.out = whole[:short]
            dec.run(_dev(st))
            with pytest.raises(wah.WahError):
                dec.status()
            got = _host(whole)
            assert bool((got[short:] == 0x5A5A5A5A).all()), (name, kw)
            if kw:
                assert bool((got == 0x5A5A5A5A).all()), (name, "two launches write nothing")
            elif name.startswith("dense") and dec.route == "one pass":  # (not with WAH_FORCE_FALLBACK=1: the no-wait route)
                assert np.array_equal(got[:short], want[:short]), (name, "one pass writes the part that fits")


def test_decode_listed_tiles_stage_only_their_words(wah, oracle):
    """A tile that decode_tile_kernel puts on its list carries 64 sums of group counts (one per 64 words), and a work item of
    the list's launch stages only the words of its own 32 segments.  The shapes that decide: a highly compressed stream (every
    tile listed, the last one partly behind the stream's end), the same with fill words of count 0 strewn in (the index-map
    route of a listed tile), classic WAH with a fill of 40 million groups (a bucket that cannot hold its sum: the tile is staged
    whole) next to ordinary listed tiles, and short fills only (tiles just above the kernel's own limit of about seven groups
    per word)."""
    rng = np.random.default_rng(3)
    n = 992 * 3000 + 7
    clustered = oracle.gen_clustered(n, 31)
    st = oracle.compress(clustered)
    cases = [("clustered", st, clustered)]
    holes = np.sort(rng.choice(len(st), 40, replace=False))
    with_empties = np.insert(st, holes, np.where(np.arange(40) % 2 == 0, 0x80000000, 0xC0000000).astype(np.uint32)).astype(np.uint32)
    cases.append(("clustered + empty fills", with_empties, clustered))
    big = np.zeros(992 * 44000, np.uint32)                      # 45 million groups of zeros ...
    big[992 * 100: 992 * 400] = oracle.gen_clustered(992 * 300, 32)  # ... around a clustered and an incompressible stretch
    big[992 * 43000: 992 * 43100] = oracle.gen_uniform(992 * 100, 33, 0.5)
    cases.append(("classic WAH, a fill of 40 M groups", _py_merge_fills(oracle.compress(big)), big))
    runs = np.repeat(rng.integers(0, 2, n // 8 + 1).astype(np.uint32) * np.uint32(0xFFFFFFFF), 8)[:n].copy()  # runs of 256 bits
    cases.append(("runs of eight words", oracle.compress(runs), runs))
    for name, stream, x in cases:
        want = oracle.decompress(stream)
        assert np.array_equal(want[: x.size], x), name
        for cap in (want.size, want.size + 1, 5 * want.size):
            dec = wah.DeviceDecompressor(len(stream), cap, one_pass=True)
            for _ in range(2):
                dec.run(_dev(stream))
                assert _route_is(dec.route, "one pass")
                assert np.array_equal(_host(dec.result()), want), (name, cap)


def test_decode_workspace_named_with_different_sizes(wah, oracle):
    """One decode workspace BUFFER, handed over with the size each stream needs (what the host entry points do with the
    buffer they keep): the areas behind the control block then lie elsewhere from call to call, and nothing in them may be
    taken for an earlier call's state -- the one-pass decoder once kept the count of its deferred tiles there and walked a
    list of an earlier call's tile bases."""
    import torch

    sizes = [992 * 40, 992 * 300 + 5, 992 * 40, 2_000_000, 992 * 40, 31, 992 * 300 + 5]
    streams = [oracle.compress(oracle.gen_uniform(n, 40 + i, 0.5 if i % 2 == 0 else 0.02)) for i, n in enumerate(sizes)]
    # ... and foreign streams, whose tiles leave flag bytes and list entries behind (fill words of count 0, a fill of 3000
    # segments inside incompressible data: tiles on the deferred list), between the plain ones
    big = oracle.gen_uniform(992 * 4000, 9, 0.5)
    big[992 * 500 + 3: 992 * 3500] = 0
    holes = _py_merge_fills(oracle.compress(big))
    empties = oracle.compress(oracle.gen_uniform(992 * 120, 10, 0.1)).copy()
    empties = np.insert(empties, [5, 700, 9000, len(empties) - 1], [0x80000000, 0xC0000000, 0x80000000, 0xC0000000]).astype(np.uint32)
    sizes = sizes[:3] + [992 * 4000, 992 * 120] + sizes[3:] + [992 * 4000]
    streams = streams[:3] + [holes, empties] + streams[3:] + [holes]
    L = wah.lib()
    ws_max = max(int(L.wah_decompress_workspace_bytes(len(st), 0)) for st in streams)
    ws = torch.zeros(ws_max, dtype=torch.uint8, device="cuda")  # zeroed ONCE
    out = torch.empty(max(sizes) + 2, dtype=torch.int32, device="cuda")
    info = torch.zeros(2, dtype=torch.int64, device="cuda")
    for rnd in range(2):
        for n, st in zip(sizes, streams):
            d = _dev(st)
            ws_bytes = int(L.wah_decompress_workspace_bytes(len(st), 0))
            rc = L.wah_decompress_device(d.data_ptr(), len(st), out.data_ptr(), n + 2, info.data_ptr(), ws.data_ptr(), ws_bytes, None)
            assert rc == 0, (n, rc)
            assert L.wah_decompress_status(ws.data_ptr(), None) == 0, n
            assert np.array_equal(_host(out[:n]), oracle.decompress(st)[:n]), n


_SCAN_ROUTE_ONLY = pytest.mark.skipif(os.environ.get("WAH_FORCE_FALLBACK") == "1",
                                      reason="tests the scan route's workspace protocol (epochs, tickets); WAH_FORCE_FALLBACK=1 "
                                             "sends every launch down the no-wait route, which reads nothing of the workspace")


@_SCAN_ROUTE_ONLY
def test_launch_epoch_wraps_around(wah, oracle):
    """The epoch that stamps a launch's granules has 16 bits.  Start a workspace just below the end of the range: the
    launches that cross it (the wrapping one has tile 0 clear the scan area while the others wait) give the same
    stream as any other launch, for the compressor and for the sums kernel of the decoder."""
    import torch

    n = 992 * 20000
    data = oracle.gen_uniform(n, 5, 0.02)
    want = oracle.compress(data)
    d = _dev(data)
    comp = wah.DeviceCompressor(n)
    dec = wah.DeviceDecompressor(len(want), n + 1)
    for ws in (comp.workspace, dec.workspace):
        ctrl = ws[:1024].view(torch.int32)
        ctrl[64] = 65533            # kCtlEpoch: two launches below the wrap value
        ctrl[65] = 0x57414832       # kCtlMagic: "a launch has completed"
    stream = _dev(want)
    for _ in range(5):
        comp.run(d)
        assert np.array_equal(_host(comp.result()), want)
        dec.run(stream)
        assert np.array_equal(_host(dec.result())[:n], data)
    assert int(comp.workspace[:1024].view(torch.int32)[64].item()) in (2, 3, 4)  # wrapped, and counting again
    assert int(comp.workspace[:1024].view(torch.int32)[66].item()) == 1          # kCtlWraps


@_SCAN_ROUTE_ONLY
def test_uninitialised_workspace_is_reported(wah, oracle):
    """A workspace that is neither zeroed nor left by an earlier launch: WAH_ERR_WORKSPACE, not a wrong stream."""
    import torch

    n = 992 * 100
    data = oracle.gen_uniform(n, 9, 0.1)
    comp = wah.DeviceCompressor(n)
    comp.workspace.fill_(0x5A)
    comp.run(_dev(data))
    with pytest.raises(wah.WahError, match="not initialised"):
        comp.status()
    # wah_workspace_init_device makes it usable again
    assert wah.lib().wah_workspace_init_device(comp.workspace.data_ptr(), comp.ws_bytes, None) == 0
    comp.run(_dev(data))
    assert np.array_equal(_host(comp.result()), oracle.compress(data))
    dec = wah.DeviceDecompressor(comp.count.item(), n + 1)
    dec.workspace.fill_(0x5A)
    dec.run(comp.out)
    with pytest.raises(wah.WahError, match="not initialised"):
        dec.status()
    torch.cuda.synchronize()
    # what recycled memory usually looks like: magic word 0 ("fresh") with garbage elsewhere.  A ticket counter that does
    # not start at zero hands out tile numbers outside the grid, an epoch no launch can have left is garbage: both are
    # reported before anything is indexed with them (kCtlStart = word 0, kCtlEpoch = word 64 of the control block).
    for word, value in ((0, 0x7FFF0000), (64, 0x12345678)):
        comp2 = wah.DeviceCompressor(n)
        comp2.workspace.view(torch.int32)[word] = value
        comp2.run(_dev(data))
        with pytest.raises(wah.WahError, match="not initialised"):
            comp2.status()
        dec2 = wah.DeviceDecompressor(comp.count.item(), n + 1)
        dec2.workspace.view(torch.int32)[word] = value
        dec2.run(comp.out)
        with pytest.raises(wah.WahError, match="not initialised"):
            dec2.status()
        un = wah.DeviceCompressor(n, unsegmented=True)
        un.workspace.view(torch.int32)[word] = value
        un.run(_dev(data))
        with pytest.raises(wah.WahError, match="not initialised"):
            un.status()


def test_two_host_threads_two_streams_one_device(wah, oracle):
    """Two host threads, each with its own stream, compressor and decompressor, hammer the same GPU at once (SURVEY.md
    section 8e: one host thread per GPU stream).  The tile kernels of the two streams run side by side; each tile only
    waits for tiles that are already running (arrival tickets), so neither starves the other, and every result is
    bit-exact.  The host-pointer entry points are called from both threads as well (one buffer set per device, calls
    take turns)."""
    import threading

    import torch

    n = 992 * 40000
    inputs = [oracle.gen_uniform(n, 21, 0.01), oracle.gen_clustered(n, 22)]
    wants = [oracle.compress(x) for x in inputs]
    small = oracle.gen_uniform(992 * 50, 23, 0.2)
    small_want = oracle.compress(small)
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d = _dev(inputs[k])
                comp = wah.DeviceCompressor(n)
                dec = wah.DeviceDecompressor(len(wants[k]), n + 1)
                for it in range(12):
                    comp.run(d, stream=stream)
                    dec.run(comp.out, c_words=len(wants[k]), stream=stream)
                    if it % 4 == 3:
                        comp.status(stream)
                        dec.status(stream)
                        assert int(comp.count.item()) == len(wants[k])
                        assert np.array_equal(_host(comp.out[: len(wants[k])]), wants[k])
                        assert np.array_equal(_host(dec.out[:n]), inputs[k])
                        assert np.array_equal(wah.compress(small), small_want)
                stream.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread is stuck"
    assert not errors, errors


def test_device_api_is_graph_capturable(wah, oracle):
    """Nothing is allocated and nothing synchronises inside the device-pointer calls: a compress + decompress pair can be
    captured into a HIP graph and replayed (every replay advances the workspaces' launch epochs like a plain launch)."""
    import torch

    n = 992 * 3000 + 5
    a = _dev(oracle.gen_uniform(n, 1, 0.01))
    b = _dev(oracle.gen_clustered(n, 2))
    d_in = a.clone()
    comp = wah.DeviceCompressor(n)
    dec = wah.DeviceDecompressor(comp.capacity, n + 1)
    comp.run(d_in)  # warm-up outside the capture
    dec.run(comp.out, comp.capacity)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            comp.run(d_in)
            # the compressed length is only known on the device: decode a stream padded with empty fills to capacity
            dec.run(comp.out, comp.capacity)
    for src in (a, b, a):
        comp.out.fill_(-2147483648)  # 0x80000000: a fill of count 0, expands to nothing
        d_in.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        comp.status()
        dec.status()
        c = int(comp.count.item())
        assert np.array_equal(_host(comp.out[:c]), oracle.compress(_host(src)))
        assert bool(torch.equal(dec.out[:n], src))


def test_indexed_path_is_graph_capturable(wah, oracle):
    """Indexed compress -> index decode and indexed compress x 2 -> wah_bitop_indexed_device, captured and replayed."""
    import torch

    n = 992 * 2000
    srcs = [_dev(oracle.gen_uniform(n, 5, 0.02)), _dev(oracle.gen_clustered(n, 6, 900))]
    d_a, d_b = srcs[0].clone(), srcs[1].clone()
    ca, cb = wah.DeviceCompressor(n, indexed=True), wah.DeviceCompressor(n, indexed=True)
    ca.run(d_a)  # warm-up outside the capture
    cb.run(d_b)
    seg_ws = torch.empty(int(wah.lib().wah_decompress_segments_workspace_bytes()), dtype=torch.uint8, device="cuda:0")
    back = torch.empty(n + 1, dtype=torch.int32, device="cuda:0")
    sc = torch.empty(int(wah.lib().wah_bitop_indexed_scratch_bytes(n)), dtype=torch.uint8, device="cuda:0")
    res = torch.empty(wah.max_compressed_words(n), dtype=torch.int32, device="cuda:0")
    res_offs = torch.zeros(n // 992 + 2, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ca.run(d_a)
            cb.run(d_b)
            # the compressed lengths are only known on the device: pass the capacities, the index bounds every segment
            wah.decompress_segments_device(ca.out, ca.seg_offsets, n, out=back, workspace=seg_ws, check=False)
            _, count, _ = wah.bitop_indexed_device("xor", ca.out, ca.seg_offsets, cb.out, cb.seg_offsets, n, scratch=sc,
                                                   out=res, out_offsets=res_offs, check=False)
    for swap in (False, True, False):
        x, y = (srcs[1], srcs[0]) if swap else (srcs[0], srcs[1])
        d_a.copy_(x)
        d_b.copy_(y)
        g.replay()
        torch.cuda.synchronize()
        ca.status()
        cb.status()
        assert bool(torch.equal(back[:n], x))
        want = oracle.compress(_host(x) ^ _host(y))
        assert np.array_equal(_host(res[: int(count.item())]), want)


def test_misaligned_input_pointer(wah, oracle):
    """A device pointer that is only 4-byte aligned takes the scalar staging path: same words."""
    n = 992 * 50 + 3
    data = oracle.gen_uniform(n + 1, 21, 0.05)
    d = _dev(data)
    got = _host(wah.compress_device(d[1:]))
    assert np.array_equal(got, oracle.compress(data[1:]))
    want = oracle.compress(data)
    padded = np.concatenate([np.zeros(1, np.uint32), want])
    back = _host(wah.decompress_device(_dev(padded)[1:], n + 2))
    assert np.array_equal(back[: n + 1], data)


def test_capacity_and_workspace_errors(wah, oracle):
    import ctypes

    import torch

    lib = wah.lib()
    n = 992 * 64
    data = oracle.gen_uniform(n, 1, 0.5)
    d_in = _dev(data)
    comp = wah.DeviceCompressor(n)
    # too small a workspace is refused before anything is launched
    rc = lib.wah_compress_device(d_in.data_ptr(), n, comp.out.data_ptr(), comp.capacity, comp.count.data_ptr(),
                                 comp.workspace.data_ptr(), 16, None)
    assert rc == -2
    # an output capacity below C is reported through the status word, nothing is written past it
    guard = torch.full((n,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    rc = lib.wah_compress_device(d_in.data_ptr(), n, guard.data_ptr(), 1000, comp.count.data_ptr(),
                                 comp.workspace.data_ptr(), comp.ws_bytes, None)
    assert rc == 0
    assert lib.wah_compress_status(comp.workspace.data_ptr(), None) == -4
    assert bool((guard[1000:] == 0x5A5A5A5A).all())
    # decoder: capacity below the decoded size
    want = oracle.compress(data)
    dec = wah.DeviceDecompressor(len(want), 100)
    dec.run(_dev(want))
    with pytest.raises(wah.WahError):
        dec.status()
    assert ctypes.c_char_p(lib.wah_last_error()).value


# ---------------------------------------------------------------- BASELINE full-size configurations
def _whole_stream_equals_oracle(oracle, d_in, c, slice_segments=1 << 21):
    """The ENTIRE compressed stream `c` of the bitmap `d_in` (both on the device) against the oracle on the host's cores
    ("compressed == CPU oracle"; source.cpp:97-103 compares every word).  The bitmap is taken to the
    host in slices of whole segments (2 GiB by default; segments are independent, F4 -- the oracle's own threads split
    it the same way); where a slice's words lie in the stream follows from the ORACLE's counts, not from anything the
    GPU reported.  Returns the number of words compared (= the oracle's C)."""
    threads = min(os.cpu_count() or 1, 256)
    n = d_in.numel()
    step = 992 * slice_segments
    out = np.empty(oracle.max_words(min(n, step)) + 1, np.uint32)
    pos = 0
    for lo in range(0, max(n, 1), step):
        part = _host(d_in[lo: lo + step])
        want = oracle.compress_mt_into(part, out, threads)
        got = _host(c[pos: pos + len(want)])
        assert len(got) == len(want), (lo, pos, len(got), len(want))
        if not np.array_equal(got, want):
            bad = int(np.flatnonzero(got != want)[0])
            raise AssertionError(f"compressed word {pos + bad} (slice at input word {lo}): {got[bad]:#x} != oracle {want[bad]:#x}")
        pos += len(want)
        del part, got
    assert pos == c.numel(), (pos, c.numel())
    return pos


def _full_size_case(wah, oracle, d_in, n, expect_ratio=None):
    """BASELINE sizes: the WHOLE compressed stream == the oracle's (every word; the oracle runs on all host cores), the
    segment index against the stream, decoded size, round-trip identity on the device."""
    import torch

    comp = wah.DeviceCompressor(n, indexed=True)
    comp.run(d_in)
    c = comp.result()
    C = c.numel()
    offs = comp.seg_offsets
    assert int(offs[0]) == 0 and int(offs[-1]) == C
    assert bool((offs[1:] > offs[:-1]).all())
    if expect_ratio:
        assert expect_ratio[0] < C / n < expect_ratio[1], C / n
    assert _whole_stream_equals_oracle(oracle, d_in, c) == C
    # the segment index against the oracle: where a few segments start, the last ones (tail path) among them
    n_seg = (wah.max_compressed_words(n) + 1023) // 1024
    for seg in sorted({1, n_seg // 3, n_seg - min(3, n_seg)} & set(range(n_seg))):
        want = oracle.compress(_host(d_in[992 * seg: (992 * (seg + 2) if seg + 3 < n_seg else n)]))
        at = int(offs[seg])
        assert np.array_equal(_host(c[at: at + len(want)]), want), seg
    # round trip on the device
    dec = wah.DeviceDecompressor(C, n + 1)
    dec.run(c)
    back = dec.result()
    assert back.numel() == (n if n % 31 == 0 else n + 1)
    assert bool(torch.equal(back[:n], d_in))
    assert int(dec.info[1]) == (32 * n + 30) // 31
    del comp, dec
    torch.cuda.empty_cache()
    return C


@pytest.mark.parametrize("n", [268435200, 268435456])
def test_config2_sparse_1gib_round_trip(wah, oracle, n):
    """BASELINE config 2: 1 GiB uniform p=0.01, compress + decompress on one GPU (whole blocks and tail)."""
    d_in = wah.gen_uniform_device(n, 1337, 0.01)
    _full_size_case(wah, oracle, d_in, n, expect_ratio=(0.47, 0.49))


def test_config3_clustered_1gib(wah, oracle):
    """BASELINE config 3: 1 GiB clustered runs (mean 4096 bits): long-fill stress."""
    n = 268435200
    d_in = wah.gen_clustered_device(n, 1337)
    _full_size_case(wah, oracle, d_in, n, expect_ratio=(0.012, 0.022))


def test_config4_dense_1gib(wah, oracle):
    """BASELINE config 4: 1 GiB p=0.5: all literals, C = G."""
    n = 268435200
    d_in = wah.gen_uniform_device(n, 1337, 0.5)
    C = _full_size_case(wah, oracle, d_in, n)
    assert abs(C - (32 * n + 30) // 31) <= 2


def test_beyond_2_31_words(wah, oracle):
    """the reference is limited to dataSize < 2^31 words (int indices).  Here sizes, offsets and word
    indices are 64-bit: an 8 GiB + bitmap (2^31 + 992 * 5 + 3 words) round-trips, and its whole stream matches the oracle."""
    n = (1 << 31) + 992 * 5 + 3
    d_in = wah.gen_uniform_device(n, 4242, 0.01)
    _full_size_case(wah, oracle, d_in, n, expect_ratio=(0.47, 0.49))


def test_config5_columns(wah, oracle):
    """BASELINE config 5 shape on one GPU: independent 128 MiB columns through one reusable compressor."""
    import torch

    n = 33554400
    comp = wah.DeviceCompressor(n)
    for col, (kind, seed) in enumerate([("u", 11), ("c", 12), ("d", 13)]):
        d = {"u": lambda: wah.gen_uniform_device(n, seed, 0.01), "c": lambda: wah.gen_clustered_device(n, seed),
             "d": lambda: wah.gen_uniform_device(n, seed, 0.5)}[kind]()
        comp.run(d)
        c = comp.result()
        m = 992 * 1024
        want = oracle.compress(_host(d[:m]))
        assert np.array_equal(_host(c[: len(want)]), want), col
        dec = wah.DeviceDecompressor(c.numel(), n + 1)
        dec.run(c)
        assert bool(torch.equal(dec.result()[:n], d)), col
        del dec


def test_column_matrix_one_launch(wah, oracle):
    """Many equal-length columns (whole 992-word segments) in ONE launch: the stream is the columns' streams back to
    back, each bit-identical to compressing that column alone, and it decodes back to the matrix."""
    import torch

    n = 992 * 700
    specs = [wah.columns.column_spec(c, n, seed=50) for c in range(7)]
    matrix = wah.columns.make_column_matrix(wah, specs, "cuda:0")
    comp = wah.DeviceCompressor(matrix.numel(), indexed=True)
    stream, offs = wah.columns.compress_column_matrix(comp, matrix)
    offs = offs.cpu().numpy()
    assert offs[0] == 0 and offs[-1] == stream.numel() and len(offs) == len(specs) + 1
    one = wah.DeviceCompressor(n)
    for c, sp in enumerate(specs):
        col = wah.columns.make_column(wah, sp, "cuda:0")
        assert bool(torch.equal(col, matrix[c]))
        one.run(col)
        alone = one.result()
        assert bool(torch.equal(alone, stream[offs[c]: offs[c + 1]])), c
        assert np.array_equal(_host(alone), oracle.compress(_host(col))), c
    back = wah.decompress_device(stream, matrix.numel() + 1)
    assert bool(torch.equal(back[: matrix.numel()].view(matrix.shape), matrix))
    with pytest.raises(ValueError):
        wah.columns.compress_column_matrix(comp, matrix[:, :991].contiguous())
    # one column out of the batch, through the segment index: nothing else of the stream is read
    segs = n // 992
    for c in (0, 3, 6):
        col = wah.decompress_segments_device(stream, comp.seg_offsets, matrix.numel(), c * segs, segs)
        assert bool(torch.equal(col, matrix[c])), c
    # ... and a stream that does not start at the allocation (4-byte aligned only)
    shifted = torch.empty(stream.numel() + 3, dtype=torch.int32, device="cuda:0")[3:]
    shifted.copy_(stream)
    assert bool(torch.equal(wah.decompress_segments_device(shifted, comp.seg_offsets, matrix.numel(), 2 * segs, segs), matrix[2]))


def test_column_shards_by_the_multi_device_entry(wah, oracle):
    """wah_compress_columns_multi_device: one call, one host thread per shard inside the library, each on the device the
    shard names and on a stream of its own.  On this box every shard names device 0 (two and three shards sharing it, as
    two streams would); on a node the same call spreads the shards over its GPUs.  Every column == the oracle's stream."""
    n = 992 * 500
    for n_shards in (1, 2, 3):
        shards, mats = [], []
        for s in range(n_shards):
            specs = [wah.columns.column_spec(c, n, seed=70) for c in range(s, 7, n_shards)]  # column c on shard c mod n_shards
            m = wah.columns.make_column_matrix(wah, specs, "cuda:0")
            shards.append((m, wah.DeviceCompressor(m.numel(), indexed=True)))
            mats.append((specs, m))
        res = wah.columns.compress_shards_multi_device(wah, shards)
        for (specs, m), (stream, offs) in zip(mats, res):
            offs = offs.cpu().numpy()
            assert offs[0] == 0 and offs[-1] == stream.numel() and len(offs) == len(specs) + 1
            for c in range(len(specs)):
                assert np.array_equal(_host(stream[offs[c]: offs[c + 1]]), oracle.compress(_host(m[c]))), (n_shards, specs[c].index)
    # a shard whose columns are not whole segments is refused before anything is launched
    with pytest.raises(ValueError):
        wah.columns.compress_shards_multi_device(wah, [(m[:, :991].contiguous(), shards[0][1])])


def _indexed_stream(wah, d_in):
    comp = wah.DeviceCompressor(d_in.numel(), indexed=True)
    comp.run(d_in)
    return comp.result().clone(), comp.seg_offsets.clone()


def test_tile_shapes_of_a_launch(wah, oracle):
    """compress_pair_kernel runs tiles of two shapes in one launch: whole rounds of the chip's 512 workgroup slots as tiles
    of 48 segments, what is left over as tiles of 16, 32 or 48 segments (compress_tile_shape).  Bitmaps right at the
    switches between the shapes -- one round = 12 288 pairs of segments, the tail shapes change at 4096 and 8192 pairs
    left over -- against the oracle, whole streams; the last one with a partial segment at its end."""
    import torch

    for pairs, extra in ((4096, 0), (4097, 0), (8192, 0), (8193, 0), (12288, 0), (12289, 0), (12288 + 4096, 0), (12288 + 4097, 0),
                         (12288 + 8193, 0), (2 * 12288 + 5, 0), (12288 + 100, 992 + 17)):
        n = pairs * 2 * 992 + extra
        d = wah.gen_uniform_device(n, 1000 + pairs, 0.003)
        comp = wah.DeviceCompressor(n, indexed=True)
        comp.run(d)
        got = _host(comp.result())
        want = oracle.compress(_host(d))
        assert np.array_equal(got, want), (pairs, extra)
        offs = comp.seg_offsets[: (wah.max_compressed_words(n) + 1023) // 1024 + 1]
        assert int(offs[0]) == 0 and int(offs[-1]) == len(want) and bool((offs[1:] > offs[:-1]).all())
        del comp, d
    torch.cuda.synchronize()


def _check_column_launch(wah, oracle, n_columns, n=33554400, seed=1337):
    """`n_columns` columns of `n` words (the three bench distributions in turn) compressed in ONE launch, exactly as
    bench.py's columns workload does it (column matrix resident in HBM, indexed compressor sized for the whole batch).
    Checked: the WHOLE stream of the launch == the oracle's, every column, every superrow of the scan (the columns are
    whole segments, so the launch's stream is the oracle's stream of the flat matrix; taken to the host 16 columns at a
    time); the column offsets out of the segment index == the oracle's column lengths; the index decode of ALL segments
    and the general decoder over the whole stream == the matrix; a column of each distribution == compressing it alone.
    Returns the number of words of the launch."""
    import torch

    specs = [wah.columns.column_spec(c, n, seed) for c in range(n_columns)]
    matrix = wah.columns.make_column_matrix(wah, specs, "cuda:0")
    comp = wah.DeviceCompressor(matrix.numel(), indexed=True)
    stream, offs = wah.columns.compress_column_matrix(comp, matrix)
    offs = offs.cpu().numpy()
    segs = n // 992
    assert len(offs) == n_columns + 1 and offs[0] == 0 and offs[-1] == stream.numel() == int(comp.count.item())
    every = comp.seg_offsets[: n_columns * segs + 1]
    assert bool((every[1:] > every[:-1]).all())
    # every word of the launch against the oracle, and the column offsets against the oracle's counts
    flat = matrix.view(-1)
    assert _whole_stream_equals_oracle(oracle, flat, stream, slice_segments=16 * segs) == stream.numel()
    threads = min(os.cpu_count() or 1, 256)
    buf = np.empty(oracle.max_words(n) + 1, np.uint32)
    for c in sorted(set(list(range(min(3, n_columns))) + [n_columns // 2] + list(range(max(n_columns - 3, 0), n_columns)))):
        assert offs[c + 1] - offs[c] == len(oracle.compress_mt_into(_host(matrix[c]), buf, threads)), c
    # a column of each distribution == compressing that column alone
    one = wah.DeviceCompressor(n)
    for c in range(min(3, n_columns)):
        one.run(matrix[c])
        assert bool(torch.equal(one.result(), stream[offs[c]: offs[c + 1]])), c
    del one
    # the index decode of every segment of the launch, 16 columns at a time, == the matrix
    for c0 in range(0, n_columns, 16):
        c1 = min(c0 + 16, n_columns)
        back = wah.decompress_segments_device(stream, comp.seg_offsets, matrix.numel(), first_segment=c0 * segs, n_segments=(c1 - c0) * segs)
        assert bool(torch.equal(back, matrix[c0:c1].reshape(-1))), c0
        del back
    # ... and the general decoder (no index) over the whole stream
    back = wah.decompress_device(stream, matrix.numel() + 1)
    assert bool(torch.equal(back[: matrix.numel()], flat))
    del back
    return matrix.numel()


def test_config5_one_launch_at_the_bench_shape(wah, oracle):
    """BASELINE config 5 at the shape bench.py times: 128 columns x 33 554 400 words = 4 294 963 200 words (16 GiB, 4096
    words below 2^32) in ONE launch."""
    assert _check_column_launch(wah, oracle, 128) == 4294963200


def test_one_launch_above_2_32_words(wah, oracle):
    """130 columns = 4 362 072 000 words in one launch: word indices, segment numbers and offsets beyond 32 bits
    (include/wah.h: n_words < 2^40)."""
    assert _check_column_launch(wah, oracle, 130, seed=77) > 1 << 32


@pytest.mark.parametrize("n", [1, 31, 991, 992, 993, 992 * 3 + 17, 992 * 64, 262144 + 5])
def test_decode_segments_through_the_index(wah, oracle, n):
    """wah_decompress_segments_device: with the segment index of the indexed compressor every 992-word segment decodes
    on its own.  Whole bitmap == the oracle's decode of the stream; any sub-range == that slice."""
    rng = np.random.default_rng(n)
    for kind in ("sparse", "dense", "runs"):
        if kind == "sparse":
            a = oracle.gen_uniform(n, 3 + n, 0.01)
        elif kind == "dense":
            a = oracle.gen_uniform(n, 4 + n, 0.5)
        else:
            a = oracle.gen_clustered(n, 5 + n, 300)
        stream, offs = _indexed_stream(wah, _dev(a))
        want = oracle.decompress(_host(stream))
        full = _host(wah.decompress_segments_device(stream, offs, n))
        assert np.array_equal(full, want), kind
        assert np.array_equal(full[:n], a), kind
        segs = (wah.max_compressed_words(n) + 1023) // 1024
        for _ in range(4):
            first = int(rng.integers(0, segs))
            count = int(rng.integers(0, segs - first + 1))
            part = _host(wah.decompress_segments_device(stream, offs, n, first, count))
            assert np.array_equal(part, want[first * 992: (first + count) * 992]), (kind, first, count)


def test_decode_segments_scatter_shapes(wah, oracle):
    """The index decode expands whole segments of up to 768 words by SCATTER into an image of the segment (a literal OR-ed in
    as one or two pieces, a fill of ones as its two ends + whole words), larger ones and the bitmap's last segment by gather
    (gpu-wah_amd/csrc/wah_decode.hip: seg_expand_scatter).  The shapes that decide: all ones (one fill word per segment), ones
    with single zero bits (fills of ones of every length and alignment between literals), literals and one-group fills
    alternating, a literal in a segment's last group, segments of exactly 768 and 769 words, run lengths around the group and
    the 32-bit word -- and an output buffer that is only 4-byte aligned (gather for every segment)."""
    import torch

    rng = np.random.default_rng(77)
    n = 992 * 24
    cases = {}
    cases["all ones"] = np.full(n, 0xFFFFFFFF, np.uint32)
    holes = np.full(n * 32, 1, np.uint8)
    holes[rng.integers(0, n * 32, 900)] = 0
    cases["ones with holes"] = holes
    alt = np.zeros(n * 32, np.uint8)
    for g in range(0, n * 32 // 31):
        alt[31 * g: 31 * g + 31] = 1 if g % 2 == 0 else rng.integers(0, 2, 31)
    cases["one-group fills of ones between literals"] = alt
    last = np.zeros(n * 32, np.uint8)
    last[31 * 1023 + 30::31 * 1024] = 1          # the last bit of every segment's last group
    last[31 * 1024 * 3: 31 * 1024 * 4] = 1       # a segment that is one fill of ones
    cases["a literal in the last group"] = last
    for words in (768, 769):
        # a segment of exactly `words` words: words - 1 literals, then one fill over the rest (segments 2 and 5)
        b = np.zeros(n * 32, np.uint8)
        for seg in (2, 5):
            g0 = 1024 * seg
            for g in range(words - 1):
                b[31 * (g0 + g) + (g % 31)] = 1
            b[31 * (g0 + words - 1): 31 * (g0 + 1024)] = seg == 5
        cases[f"{words} words in a segment"] = b
    runs = np.zeros(n * 32, np.uint8)
    pos, v = 0, 0
    while pos < n * 32:
        ln = int(rng.choice([1, 2, 30, 31, 32, 33, 62, 63, 64, 65, 31 * 5, 31 * 64 + 3, 31 * 700]))
        runs[pos: pos + ln] = v
        pos, v = pos + ln, 1 - v
    cases["runs around the group and the word"] = runs
    for name, c in cases.items():
        a = c if c.dtype == np.uint32 else np.packbits(c[: n * 32].reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
        stream, offs = _indexed_stream(wah, _dev(a))
        assert np.array_equal(_host(stream), oracle.compress(a)), name
        got = _host(wah.decompress_segments_device(stream, offs, n))
        assert np.array_equal(got[:n], a), name
        # the same into a buffer that begins 4 bytes behind a 16-byte boundary, and a range in the middle
        shifted = torch.empty(n + 8, dtype=torch.int32, device="cuda")[1:]
        got = _host(wah.decompress_segments_device(stream, offs, n, out=shifted))
        assert np.array_equal(got[:n], a), (name, "4-byte aligned output")
        part = _host(wah.decompress_segments_device(stream, offs, n, 3, 9))
        assert np.array_equal(part, a[3 * 992: 12 * 992]), (name, "segments 3..11")


def test_decode_segments_rejects_what_is_not_a_segmented_stream(wah, oracle):
    import torch

    n = 992 * 40
    a = oracle.gen_clustered(n, 77, 500)
    stream, offs = _indexed_stream(wah, _dev(a))
    with pytest.raises(wah.WahError):      # capacity
        wah.decompress_segments_device(stream, offs, n, out=torch.empty(n - 1, dtype=torch.int32, device="cuda:0"))
    with pytest.raises(wah.WahError):      # range outside the bitmap
        wah.decompress_segments_device(stream, offs, n, 30, 20)
    bad = offs.clone()
    bad[7] += 1                            # segment 6 gets a word too many, segment 7 one too few
    with pytest.raises(wah.WahError):
        wah.decompress_segments_device(stream, bad, n)
    bad = offs.clone()
    bad[5], bad[6] = offs[6].item(), offs[5].item()   # not monotonic
    with pytest.raises(wah.WahError):
        wah.decompress_segments_device(stream, bad, n)
    bad = offs.clone()
    bad[-1] = stream.numel() + 5           # past the stream
    with pytest.raises(wah.WahError):
        wah.decompress_segments_device(stream, bad, n)
    words = _host(stream).copy()
    fills = np.flatnonzero(words & 0x80000000)
    assert fills.size
    words[fills[0]] += 3                   # a fill that is three groups too long
    with pytest.raises(wah.WahError):
        wah.decompress_segments_device(_dev(words), offs, n)
    words = _host(stream).copy()
    words[fills[0]] &= 0xC0000000          # an empty fill
    with pytest.raises(wah.WahError):
        wah.decompress_segments_device(_dev(words), offs, n)
    # the untouched stream still decodes after all that
    assert np.array_equal(_host(wah.decompress_segments_device(stream, offs, n))[:n], a)


def test_build_index_for_streams_that_came_without_one(wah, oracle):
    """wah_build_index_device: the index of a plain compress() stream == the one the indexed compressor writes; streams
    that are not segmented (a merged fill across a boundary, an empty fill) have none."""
    for n in (1, 31, 992, 993, 992 * 3 + 17, 4096 * 31 + 5, 992 * 700):
        for kind in ("sparse", "dense", "runs"):
            a = {"sparse": lambda: oracle.gen_uniform(n, 3 + n, 0.01), "dense": lambda: oracle.gen_uniform(n, 4 + n, 0.5),
                 "runs": lambda: oracle.gen_clustered(n, 5 + n, 3000)}[kind]()
            stream = _dev(oracle.compress(a))                     # a stream from elsewhere: the CPU oracle
            offs, groups = wah.build_index_device(stream)
            _, want = _indexed_stream(wah, _dev(a))
            assert groups == wah.max_compressed_words(n)
            assert np.array_equal(offs.cpu().numpy(), want.cpu().numpy()), (n, kind)
            back = _host(wah.decompress_segments_device(stream, offs, n))
            assert np.array_equal(back[:n], a), (n, kind)
    zeros = np.zeros(992 * 5, dtype=np.uint32)
    merged = wah.merge_fills_device(_dev(oracle.compress(zeros)))  # one fill of 5120 groups
    assert merged.numel() == 1
    with pytest.raises(wah.WahError):
        wah.build_index_device(merged)
    words = oracle.compress(oracle.gen_clustered(992 * 9, 2, 500)).copy()
    words[np.flatnonzero(words & 0x80000000)[0]] &= 0xC0000000    # an empty fill
    with pytest.raises(wah.WahError):
        wah.build_index_device(_dev(words))


def test_decode_segments_full_size(wah):
    """BASELINE size (1 GiB bitmap): index decode == the bitmap, for the three bench distributions."""
    import torch

    n = 992 * 1024 * 264
    for c in range(3):
        spec = wah.columns.column_spec(c, n, seed=9)
        col = wah.columns.make_column(wah, spec, "cuda:0")
        stream, offs = _indexed_stream(wah, col)
        back = wah.decompress_segments_device(stream, offs, n)
        assert bool(torch.equal(back[:n], col)), spec.kind
        del stream, offs, back, col


def test_ragged_columns_one_launch(wah, oracle):
    """Columns of different lengths (multiples of 992 words) back to back in one buffer: one launch, every column's
    stream bit-identical to compressing it alone; one column decoded through the index."""
    import torch

    lengths = [992 * 3, 992 * 700, 992, 992 * 41, 992 * 256]
    cols = [oracle.gen_uniform(n, 60 + i, (0.01, 0.5, 0.2)[i % 3]) if i % 2 == 0 else oracle.gen_clustered(n, 60 + i, 700)
            for i, n in enumerate(lengths)]
    flat = _dev(np.concatenate(cols))
    comp = wah.DeviceCompressor(flat.numel(), indexed=True)
    stream, offs = wah.columns.compress_column_ranges(comp, flat, lengths)
    offs = offs.cpu().numpy()
    assert offs[0] == 0 and offs[-1] == stream.numel() and len(offs) == len(lengths) + 1
    for c, col in enumerate(cols):
        assert np.array_equal(_host(stream[offs[c]: offs[c + 1]]), oracle.compress(col)), c
    first = sum(lengths[:3]) // 992
    back = wah.decompress_segments_device(stream, comp.seg_offsets, flat.numel(), first, lengths[3] // 992)
    assert np.array_equal(_host(back), cols[3])
    with pytest.raises(ValueError):
        wah.columns.compress_column_ranges(comp, flat, [992 * 3 + 1] + lengths[1:])


# ---------------------------------------------------------------- the reference's own test code
def test_reference_tests_cpp_against_hip_library(wah):
    """oracle/_ref/ref_tests_hip = /root/reference/tests.cpp compiled in the authoring container and linked
    against libwah_hip.so: the reference's own callers drive the drop-in boundary on the GPU.  Expected:
    10 pass, the 2 stale-vector tests and the obsolete extendDataTest fail (exit code 0 iff exactly that)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_tests_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_tests_hip was not built (no /root/reference in this checkout)")
    r = subprocess.run([exe, "--big"], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("[ref-test]")]
    assert r.returncode == 0, "\n".join(lines) + r.stderr[-2000:]
    assert sum("PASS" in ln for ln in lines) == 10, "\n".join(lines)
