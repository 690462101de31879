#!/usr/bin/env python3
"""Per-tile timeline of the compress kernel from the DIAGNOSTIC build: when each tile's iteration started, when it
was classified and when its offset was resolved (10 ns ticks).  Prints summary statistics over tile index."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "gpu-wah_amd", "libwah_hip_diag.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
n = 268435200
d = wah.gen_uniform_device(n, 1337, 0.01)
comp = wah.DeviceCompressor(n, indexed=True)
comp.run(d)
comp.status()
comp.seg_offsets.zero_()
comp.run(d)
comp.status()
tl = comp.seg_offsets.cpu().numpy()
T = (comp.capacity + 1023) // 1024
T = (T + 6) // 7
tl = tl[: 4 * T].reshape(T, 4).astype(np.int64)
t0 = tl[:, 0].min()
start, cls, res = [(tl[:, i] - t0) / 100.0 for i in range(3)]
wg = tl[:, 3] >> 32
it = tl[:, 3] & 0xFFFFFFFF
print(f"tiles {T}, kernel span {res.max():.1f} us")
print(f"classify time (start->classified): mean {np.mean(cls - start):.2f} us, p50 {np.median(cls - start):.2f}, p99 {np.percentile(cls - start, 99):.2f}")
print(f"look-back wait (classified->resolved): mean {np.mean(res - cls):.2f} us, p50 {np.median(res - cls):.2f}, p99 {np.percentile(res - cls, 99):.2f}")
for lo in (0, 8, 16, 64, 768, 1536, 2304, 5000, 20000, 38000):
    idx = np.arange(lo, min(lo + 10, T))
    print(f"tile {lo:6d}.. : " + " ".join(f"[it{it[i]} wg{wg[i]:3d} s{start[i]:7.1f} c{cls[i]:7.1f} r{res[i]:7.1f}]" for i in idx[:5]))
# how is resolved time ordered in tile index?
order = np.argsort(res)
print("resolved-time monotone in tile index? fraction of adjacent inversions:", np.mean(np.diff(res) < 0))
print("iterations per WG:", it.max() + 1, " mean iteration period (us):", res.max() / (it.max() + 1))
for g in range(0, int(it.max()) + 1, 5):
    m = it == g
    print(f"  iteration {g:3d}: tiles {m.sum():4d}  tile idx [{np.where(m)[0].min():6d},{np.where(m)[0].max():6d}]  start [{start[m].min():8.1f},{start[m].max():8.1f}]  resolved [{res[m].min():8.1f},{res[m].max():8.1f}]")
