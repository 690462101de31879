"""ctypes bindings of the CPU oracle (oracle/libwah_oracle.so) for the tests.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libwah_oracle.so")

_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64 = ctypes.c_uint64


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("wah_oracle.c", "wah_refsim.c", "wah_gen_host.c", "wah_oracle.h")]
    srcs.append(os.path.join(ROOT, "include", "wah_gen.h"))
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libwah_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB


def _ptr(a):
    return a.ctypes.data_as(_u32p)


class Oracle:
    def __init__(self, path):
        self.lib = lib = ctypes.CDLL(path)
        lib.wah_oracle_max_words.restype = _u64
        lib.wah_oracle_max_words.argtypes = [_u64]
        for name in ("wah_oracle_compress", "wah_refsim_compress", "wah_refsim_compress_pre195"):
            fn = getattr(lib, name)
            fn.restype = _u64
            fn.argtypes = [_u32p, _u64, _u32p]
        lib.wah_oracle_compress_mt.restype = _u64
        lib.wah_oracle_compress_mt.argtypes = [_u32p, _u64, _u32p, ctypes.c_int]
        lib.wah_oracle_decoded_groups.restype = _u64
        lib.wah_oracle_decoded_groups.argtypes = [_u32p, _u64]
        lib.wah_oracle_decoded_words.restype = _u64
        lib.wah_oracle_decoded_words.argtypes = [_u64]
        lib.wah_oracle_decompress.restype = _u64
        lib.wah_oracle_decompress.argtypes = [_u32p, _u64, _u32p]
        lib.wah_oracle_group.restype = ctypes.c_uint32
        lib.wah_oracle_group.argtypes = [_u32p, _u64, _u64]
        for name in ("wah_gen_host_uniform", "wah_gen_host_clustered"):
            fn = getattr(lib, name)
            fn.restype = None
            fn.argtypes = [_u32p, _u64, _u64, _u64]

    # --- format helpers
    def max_words(self, n):
        return int(self.lib.wah_oracle_max_words(n))

    @staticmethod
    def _in(data):
        a = np.ascontiguousarray(data, dtype=np.uint32)
        return a if a.size else np.zeros(1, np.uint32)[:0].copy()

    def _run_compress(self, fn, data, *extra):
        a = np.ascontiguousarray(data, dtype=np.uint32)
        out = np.empty(self.max_words(a.size) + 1, np.uint32)
        src = a if a.size else np.zeros(1, np.uint32)
        c = int(fn(_ptr(src), a.size, _ptr(out), *extra))
        assert c != 2**64 - 1, "refsim is only defined for n % 992 == 0"
        return out[:c].copy()

    def compress(self, data):
        return self._run_compress(self.lib.wah_oracle_compress, data)

    def compress_mt(self, data, threads):
        return self._run_compress(self.lib.wah_oracle_compress_mt, data, int(threads))

    def compress_mt_into(self, data, out, threads):
        """compress_mt into a caller's buffer (>= max_words(len(data)) + 1 words): no copy of the result -- for the
        whole-stream comparisons at the BASELINE sizes, where the stream is gigabytes.  Returns a view of `out`."""
        a = np.ascontiguousarray(data, dtype=np.uint32)
        assert out.dtype == np.uint32 and out.size >= self.max_words(a.size) + 1
        src = a if a.size else np.zeros(1, np.uint32)
        c = int(self.lib.wah_oracle_compress_mt(_ptr(src), a.size, _ptr(out), int(threads)))
        return out[:c]

    def refsim_compress(self, data):
        return self._run_compress(self.lib.wah_refsim_compress, data)

    def refsim_compress_pre195(self, data):
        return self._run_compress(self.lib.wah_refsim_compress_pre195, data)

    def decoded_groups(self, comp):
        c = np.ascontiguousarray(comp, dtype=np.uint32)
        src = c if c.size else np.zeros(1, np.uint32)
        return int(self.lib.wah_oracle_decoded_groups(_ptr(src), c.size))

    def decoded_words(self, groups):
        return int(self.lib.wah_oracle_decoded_words(groups))

    def decompress(self, comp):
        c = np.ascontiguousarray(comp, dtype=np.uint32)
        src = c if c.size else np.zeros(1, np.uint32)
        n = self.decoded_words(self.decoded_groups(c))
        out = np.zeros(n + 1, np.uint32)
        got = int(self.lib.wah_oracle_decompress(_ptr(src), c.size, _ptr(out)))
        assert got == n
        return out[:n].copy()

    def group(self, data, g):
        a = np.ascontiguousarray(data, dtype=np.uint32)
        return int(self.lib.wah_oracle_group(_ptr(a), a.size, g))

    def time_round_trip(self, data, threads=1, reps=2):
        """Best-of-`reps` wall times (compress_s, decompress_s) on pre-touched buffers (CPU baseline leg)."""
        import time

        a = np.ascontiguousarray(data, dtype=np.uint32)
        out = np.zeros(self.max_words(a.size) + 1, np.uint32)
        out.fill(1)  # touch every page: first-touch faults are not part of the algorithm
        back = np.zeros(a.size + 2, np.uint32)
        back.fill(1)
        best_c = best_d = float("inf")
        for _ in range(reps + (1 if threads > 1 else 0)):  # (the first multi-threaded call starts the pool's threads)
            t0 = time.perf_counter()
            if threads > 1:
                c = int(self.lib.wah_oracle_compress_mt(_ptr(a), a.size, _ptr(out), int(threads)))
            else:
                c = int(self.lib.wah_oracle_compress(_ptr(a), a.size, _ptr(out)))
            t1 = time.perf_counter()
            n = int(self.lib.wah_oracle_decompress(_ptr(out), c, _ptr(back)))
            t2 = time.perf_counter()
            best_c = min(best_c, t1 - t0)
            best_d = min(best_d, t2 - t1)
        assert np.array_equal(back[: a.size], a) and n >= a.size
        return best_c, best_d, c

    # --- generators (same bits as the HIP generator kernels; include/wah_gen.h)
    def gen_uniform(self, n_words, seed, p):
        out = np.empty(max(n_words, 1), np.uint32)
        self.lib.wah_gen_host_uniform(_ptr(out), n_words, seed, threshold_for(p))
        return out[:n_words]

    def gen_clustered(self, n_words, seed, mean_run_bits=4096):
        out = np.empty(max(n_words, 1), np.uint32)
        self.lib.wah_gen_host_clustered(_ptr(out), n_words, seed, threshold_for(1.0 / mean_run_bits))
        return out[:n_words]


def threshold_for(p):
    return min(int(p * 2**32), 2**32)


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(build())
    return _cached


def load_kats():
    with open(os.path.join(ROOT, "tests", "golden", "kats.json")) as f:
        doc = json.load(f)
    out = []
    for k in doc["kats"]:
        data = np.zeros(k["n_words"], np.uint32)
        for i, v in k["input"].items():
            data[int(i)] = v
        k = dict(k)
        k["data"] = data
        k["expected"] = np.array(k["expected"], np.uint32)
        if "stale_expected" in k:
            k["stale_expected"] = np.array(k["stale_expected"], np.uint32)
        out.append(k)
    return out


# ---- a tiny independent pure-Python statement of the format (small cases only) ----
def py_compress(data):
    """Bit-at-a-time WAH31 with 1024-group segments; O(bits) Python, for n <= a few thousand words."""
    data = [int(x) for x in data]
    nbits = 32 * len(data)
    G = (nbits + 30) // 31

    def bit(k):
        return (data[k >> 5] >> (k & 31)) & 1 if k < nbits else 0

    out = []
    run_kind, run_len = None, 0

    def flush():
        nonlocal run_kind, run_len
        if run_kind is not None:
            out.append((0xC0000000 if run_kind else 0x80000000) | run_len)
        run_kind, run_len = None, 0

    for g in range(G):
        if g % 1024 == 0:
            flush()
        x = 0
        for j in range(31):
            x |= bit(31 * g + j) << j
        if x == 0 or x == 0x7FFFFFFF:
            kind = 1 if x else 0
            if run_kind == kind:
                run_len += 1
            else:
                flush()
                run_kind, run_len = kind, 1
        else:
            flush()
            out.append(x)
    flush()
    return np.array(out, np.uint32)


def py_decompress(comp):
    bits = []
    for w in (int(x) for x in comp):
        if w & 0x80000000:
            v = 0x7FFFFFFF if (w & 0x40000000) else 0
            reps = w & 0x3FFFFFFF
        else:
            v, reps = w, 1
        for _ in range(reps):
            bits.extend((v >> j) & 1 for j in range(31))
    n = (len(bits) + 31) // 32
    bits.extend([0] * (32 * n - len(bits)))
    out = np.zeros(n, np.uint32)
    for i in range(n):
        x = 0
        for j in range(32):
            x |= bits[32 * i + j] << j
        out[i] = x
    return out
