#!/usr/bin/env python3
"""Soak run: many round trips at varying sizes and densities with every status checked (persistent kernels: a lost
workgroup or an expired wait shows up here as WAH_ERR_TIMEOUT).  usage: python tools/soak_gpu.py [iterations]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

wah = importlib.import_module("gpu-wah_amd")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(123)
t0 = time.time()
worst = 0.0
for it in range(iters):
    n = int(rng.choice([992 * 3, 992 * 15 * 256 - 5, 992 * 15 * 257 + 1, 33554400, 67108864 + 7, 268435200]))
    kind = int(rng.integers(0, 4))
    d = (wah.gen_uniform_device(n, it, float(rng.choice([0.5, 0.1, 0.01, 2.0**-12]))) if kind < 3 else wah.gen_clustered_device(n, it, int(rng.choice([64, 4096, 1 << 18]))))
    comp = wah.DeviceCompressor(n)
    ts = time.time()
    for _ in range(3):
        comp.run(d)
    c = comp.result()
    dec = wah.DeviceDecompressor(c.numel(), n + 1)
    for _ in range(3):
        dec.run(c)
    back = dec.result()
    assert back.numel() == (n if n % 31 == 0 else n + 1) and bool(torch.equal(back[:n], d)), (it, n, kind)
    worst = max(worst, time.time() - ts)
    del comp, dec, d, c, back
    if it % 25 == 0:
        print(f"iteration {it}: ok, {time.time() - t0:.0f} s, slowest case so far {worst * 1e3:.0f} ms", flush=True)
print("soak done:", iters, "iterations, no error")
