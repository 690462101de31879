#!/usr/bin/env python3
"""Time line of the compress tile kernel (DIAGNOSTIC build, WAH_TUNE=77): per tile, when it started, published its
count, got its sweep back and knew its offset (s_memrealtime, 100 MHz).  Answers: do tiles start in blockIdx order
across the XCDs, how long does a sweep take, and how far behind a tile's own count are the counts it waits for."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "gpu-wah_amd", "libwah_hip_diag.so")
os.environ.setdefault("WAH_TUNE", "78")  # 77: wave 0 waits for its first sweep right away (its round trip); 78: normal flow
import numpy as np  # noqa: E402
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
n = 268435200
for kind in sys.argv[1:] or ["sparse"]:
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n, indexed=True)
    comp.run(d)
    comp.status()
    comp.run(d)
    comp.status()
    n_tiles = (270600 + 39) // 40
    t = comp.seg_offsets[: n_tiles * 8].cpu().numpy().reshape(n_tiles, 8).astype(np.int64)
    start, pub, sweep, done, polls, p2, bar2 = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4], t[:, 5], t[:, 6]
    t0 = start.min()
    us = lambda x: x / 100.0
    print(f"--- {kind}: {n_tiles} tiles, span {us(done.max() - t0):.1f} us")
    print(f"   start->publish {us((pub - start).mean()):.2f} us (p10 {us(np.percentile(pub - start, 10)):.2f}, p90 {us(np.percentile(pub - start, 90)):.2f})")
    if os.environ["WAH_TUNE"] == "77":
        print(f"   sweep round trip {us((sweep - pub).mean()):.2f} us (p10 {us(np.percentile(sweep - pub, 10)):.2f}, p90 {us(np.percentile(sweep - pub, 90)):.2f})")
    print(f"   publish->pass 2 + final words done {us((p2 - pub).mean()):.2f} us; ->offset known {us((done - pub).mean()):.2f}; ->barrier 2 passed {us((bar2 - pub).mean()):.2f}")
    print(f"   publish->offset {us((done - pub).mean()):.2f} us (p50 {us(np.percentile(done - pub, 50)):.2f}, p90 {us(np.percentile(done - pub, 90)):.2f}); re-polls {polls.mean():.2f}")
    # start order: how much later than tile t did the latest-starting lower tile (within 512) start?
    lag_start = np.zeros(n_tiles)
    lag_pub = np.zeros(n_tiles)
    who = np.zeros(n_tiles, dtype=np.int64)
    for i in range(1, n_tiles):
        lo = max(0, i - 511)
        lag_start[i] = start[lo:i].max() - start[i]
        k = pub[lo:i].argmax()
        lag_pub[i] = pub[lo + k] - pub[i]
        who[i] = i - (lo + k)
    print(f"   latest START among the 511 tiles before a tile, relative to its own start: mean {us(lag_start.mean()):.2f} us, p90 {us(np.percentile(lag_start, 90)):.2f}, max {us(lag_start.max()):.2f}")
    print(f"   latest PUBLISH among them, relative to its own publish: mean {us(lag_pub.mean()):.2f} us, p50 {us(np.percentile(lag_pub, 50)):.2f}, p90 {us(np.percentile(lag_pub, 90)):.2f}")
    print(f"   distance to that latest publisher: p50 {np.percentile(who, 50):.0f}, p90 {np.percentile(who, 90):.0f} tiles; same XCD: {(who % 8 == 0).mean() * 100:.0f} %")
    per_xcd = [us((pub - start)[x::8].mean()) for x in range(8)]
    print("   start->publish by XCD: " + " ".join(f"{v:.2f}" for v in per_xcd))
    order = np.argsort(start, kind="stable")
    print(f"   tiles whose start rank differs from their id by more than 96: {(np.abs(order - np.arange(n_tiles)) > 96).mean() * 100:.1f} %")
    del comp, d
