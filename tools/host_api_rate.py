#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-ABI entry points (host pointers in, malloc'd host buffer out)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
wah = importlib.import_module("gpu-wah_amd")
from tests import _oracle
o = _oracle.load()
n = 992 * 1024 * 64  # 248 MiB, the reference's compressAndDecompressTest size (tests.cpp:243-244)
data = o.gen_uniform(n, 1337, 0.01)
for rep in range(3):
    t0 = time.perf_counter(); c, tc = wah.compress(data, with_timings=True); t1 = time.perf_counter()
    d, td = wah.decompress(c, with_timings=True); t2 = time.perf_counter()
    assert np.array_equal(d[:n], data)
    print(f"rep {rep}: compress() H2D {tc.to_device_ms:.2f} ms, device {tc.device_ms:.3f} ms, D2H+free {tc.from_device_ms:.2f} ms, wall {1e3*(t1-t0):.1f} ms -> {4*n/(t1-t0)/1e9:.2f} GB/s end to end; "
          f"decompress() H2D {td.to_device_ms:.2f}, device {td.device_ms:.3f}, D2H {td.from_device_ms:.2f}, wall {1e3*(t2-t1):.1f} ms -> {4*n/(t2-t1)/1e9:.2f} GB/s")
