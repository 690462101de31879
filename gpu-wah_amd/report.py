"""Benchmark / report driver: the counterpart of the reference's main() (/root/reference/source.cpp:29-148).

Same sweep -- bitmap sizes s * 1024 blocks of 31*32 words for s = 1, 2, 4 .. 256 (source.cpp:54,67), densities
"one bit in 2^i" for i = 1 .. 16 (source.cpp:57,78), ten repetitions averaged (source.cpp:70,133) -- through the same
host-pointer boundary (compress() / decompress() with the three millisecond timings), the round trip checked on every
repetition (source.cpp:103), and the SAME eleven CSV columns in the same order and wording (source.cpp:38-48), so a
results file of this driver can be laid next to a results.txt written by the reference on other hardware.  Six columns
are appended: GB/s of input bits for both device phases, the HBM roofline fraction of both (algorithmic bytes
4N + 4C and 4C + 4 ceil(31G/32) over the device time, peak 8 TB/s), and median device times.

The bitmaps come from this package's counter-based generator (include/wah_gen.h, bit set with probability 2^-i), not
from the reference's rand() loop (tests.cpp:42-64), whose output depends on the C library: sizes and densities match,
the individual bits do not.

Everything is computed by libwah_hip.so; there is no CPU path.
"""
import argparse
import statistics
import sys

import numpy as np

from . import api

# source.cpp:38-48, verbatim column titles (including their spacing)
REFERENCE_COLUMNS = [
    "Original size [Int] ",
    " Compressed size [Int] ",
    " Decompressed size [Int] ",
    " Density",
    " Compression Ratio",
    " Compression transfer to device [ms]",
    " Compression time [ms]",
    " Compression transfer from device [ms]",
    " Decompression transfer to device [ms]",
    " Decompression time [ms]",
    "Decompression transfer from device [ms]",
]
EXTRA_COLUMNS = [
    " Compression [GB/s of input]",
    " Decompression [GB/s of input]",
    " Compression roofline fraction",
    " Decompression roofline fraction",
    " Compression time median [ms]",
    " Decompression time median [ms]",
]
HBM_PEAK_BYTES_PER_S = 8.0e12
BLOCK_WORDS = 31 * 32  # one reference block (source.cpp:67)


def header(extra=True):
    return ",".join(REFERENCE_COLUMNS + (EXTRA_COLUMNS if extra else []))


def sweep_sizes(max_s=256):
    """source.cpp:54: s = 1, 2, 4, ..., 256; dataSize = s * 1024 * 31 * 32 words (source.cpp:67)."""
    s = 1
    while s <= max_s:
        yield s, s * 1024 * BLOCK_WORDS
        s <<= 1


def make_bitmap(n_words, density_exp, seed=1337):
    """Host bitmap with every bit set with probability 2^-density_exp (generated on the device, copied back)."""
    d = api.gen_uniform_device(n_words, seed, 2.0 ** -density_exp)
    return d.cpu().numpy().view(np.uint32)


def measure(data, reps):
    """One row of the report: `reps` round trips through the host boundary (source.cpp:83-127)."""
    n = int(data.size)
    c_t = [[], [], []]
    d_t = [[], [], []]
    c_words = dec_words = 0
    for _ in range(reps):
        comp, ct = api.compress(data, with_timings=True)
        back, dt = api.decompress(comp, with_timings=True)
        if not np.array_equal(back[:n], data):  # source.cpp:103
            raise api.WahError("round trip mismatch")
        c_words, dec_words = int(comp.size), int(back.size)
        for acc, t in ((c_t, ct), (d_t, dt)):
            acc[0].append(t.to_device_ms)
            acc[1].append(t.device_ms)
            acc[2].append(t.from_device_ms)
    mean = statistics.fmean
    groups = (32 * n + 30) // 31
    c_ms, d_ms = mean(c_t[1]), mean(d_t[1])
    in_bytes = 4.0 * n
    algo_c = 4.0 * n + 4.0 * c_words
    algo_d = 4.0 * c_words + 4.0 * ((31 * groups + 31) // 32)
    return {
        "n": n, "c": c_words, "d": dec_words, "ratio": c_words / n,
        "c_to": mean(c_t[0]), "c_ms": c_ms, "c_from": mean(c_t[2]),
        "d_to": mean(d_t[0]), "d_ms": d_ms, "d_from": mean(d_t[2]),
        "c_gbps": in_bytes / (c_ms * 1e-3) / 1e9, "d_gbps": in_bytes / (d_ms * 1e-3) / 1e9,
        "c_frac": algo_c / (c_ms * 1e-3) / HBM_PEAK_BYTES_PER_S, "d_frac": algo_d / (d_ms * 1e-3) / HBM_PEAK_BYTES_PER_S,
        "c_med": statistics.median(c_t[1]), "d_med": statistics.median(d_t[1]),
    }


def format_row(r, density_exp, extra=True):
    """source.cpp:129-140: same fields, same order."""
    cells = [str(r["n"]), f" {r['c']}", f" {r['d']}", f" {density_exp}", f" {r['ratio']:.6g}",
             f" {r['c_to']:.6g}", f" {r['c_ms']:.6g}", f" {r['c_from']:.6g}",
             f" {r['d_to']:.6g}", f" {r['d_ms']:.6g}", f" {r['d_from']:.6g}"]
    if extra:
        cells += [f" {r['c_gbps']:.2f}", f" {r['d_gbps']:.2f}", f" {r['c_frac']:.4f}", f" {r['d_frac']:.4f}",
                  f" {r['c_med']:.6g}", f" {r['d_med']:.6g}"]
    return ",".join(cells)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default="results.txt", help="report file, appended to like the reference's (source.cpp:36)")
    ap.add_argument("--max-s", type=int, default=256, help="largest size factor s (source.cpp:54)")
    ap.add_argument("--min-s", type=int, default=1)
    ap.add_argument("--densities", default="1-16", help="range or list of exponents i: one bit in 2^i (source.cpp:57)")
    ap.add_argument("--reps", type=int, default=10, help="repetitions per row (source.cpp:70)")
    ap.add_argument("--reference-columns-only", action="store_true", help="write exactly the reference's 11 columns")
    ap.add_argument("--seed", type=int, default=1337)
    return ap.parse_args(argv)


def parse_densities(spec):
    out = []
    for part in spec.split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    if not out or min(out) < 1 or max(out) > 32:
        raise ValueError("densities must be exponents in 1..32")
    return out


def main(argv=None):
    args = parse_args(argv)
    api.lib()  # fail now if the HIP library is missing
    extra = not args.reference_columns_only
    with open(args.out, "a") as fs:
        fs.write(header(extra) + "\n")
        for s, n_words in sweep_sizes(args.max_s):
            if s < args.min_s:
                continue
            for i in parse_densities(args.densities):
                data = make_bitmap(n_words, i, args.seed)
                row = measure(data, args.reps)
                line = format_row(row, i, extra)
                fs.write(line + "\n")
                fs.flush()
                print(f"s={s} i={i}: data matches; {line}", file=sys.stderr)
    return 0
