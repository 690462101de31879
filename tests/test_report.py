"""The report driver mirrors the reference's main() (source.cpp:29-148): columns, sweep, row format."""
import importlib
import os

import numpy as np
import pytest

wah = importlib.import_module("gpu-wah_amd")
report = wah.report


def test_header_is_the_references():
    # source.cpp:38-48 writes these eleven titles, each followed by ", " except the last two (",", endl)
    ref = ("Original size [Int] , " "Compressed size [Int] , " "Decompressed size [Int] , " "Density, " "Compression Ratio, "
           "Compression transfer to device [ms], " "Compression time [ms], " "Compression transfer from device [ms], "
           "Decompression transfer to device [ms], " "Decompression time [ms]," "Decompression transfer from device [ms]")
    assert report.header(extra=False) == ref
    assert report.header(extra=True).startswith(ref + ",")
    assert len(report.header(extra=True).split(",")) == 17


def test_sweep_matches_source_cpp():
    sizes = list(report.sweep_sizes())
    assert [s for s, _ in sizes] == [1, 2, 4, 8, 16, 32, 64, 128, 256]  # source.cpp:54
    assert sizes[0][1] == 1024 * 31 * 32 and sizes[-1][1] == 256 * 1024 * 31 * 32  # source.cpp:67
    assert all(n % 992 == 0 for _, n in sizes)  # whole reference blocks only
    assert report.parse_densities("1-16") == list(range(1, 17))  # source.cpp:57
    assert report.parse_densities("1,4,8") == [1, 4, 8]
    with pytest.raises(ValueError):
        report.parse_densities("0-3")


def test_row_format():
    r = {"n": 1015808, "c": 500000, "d": 1015808, "ratio": 0.4922, "c_to": 1.0, "c_ms": 0.5, "c_from": 2.0, "d_to": 0.7,
         "d_ms": 0.6, "d_from": 1.5, "c_gbps": 8126.46, "d_gbps": 6772.05, "c_frac": 0.6, "d_frac": 0.5, "c_med": 0.5, "d_med": 0.6}
    cells = report.format_row(r, 4, extra=False).split(",")
    assert len(cells) == 11 and cells[0] == "1015808" and cells[3].strip() == "4"
    assert len(report.format_row(r, 4).split(",")) == 17


@pytest.mark.gpu
def test_small_sweep_on_gpu(tmp_path):
    out = tmp_path / "results.txt"
    assert report.main(["--out", str(out), "--max-s", "2", "--densities", "1,6", "--reps", "2"]) == 0
    lines = out.read_text().strip().split("\n")
    assert lines[0] == report.header(True) and len(lines) == 1 + 2 * 2
    for line in lines[1:]:
        cells = [c.strip() for c in line.split(",")]
        n, c, d = int(cells[0]), int(cells[1]), int(cells[2])
        assert d == n and 0 < c <= (32 * n + 30) // 31  # whole blocks: the decoder returns exactly N words
        assert abs(float(cells[4]) - c / n) < 1e-4
        assert float(cells[6]) > 0 and float(cells[9]) > 0
    # density 1 (p = 0.5) is incompressible, density 6 compresses
    rows = [l.split(",") for l in lines[1:]]
    assert float(rows[0][4]) > 1.03 and float(rows[1][4]) < 0.9
