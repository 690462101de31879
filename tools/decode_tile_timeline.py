#!/usr/bin/env python3
"""Time line of decode_tile_kernel (DIAGNOSTIC build: the stamps overwrite the head of the output buffer): per workgroup = per
BATCH of tiles (two tiles of 8192 words; WAH_DT_BATCH=1: one), when it started, had its last tile staged and all of them counted,
published its total, had the first tile's start flags, knew its base, finished (s_memrealtime, 100 MHz).
A workload `uI` is the uniform bitmap with one bit in 2^I (u9, u10: every tile goes onto the list; the list's launch is left out --
WAH_DIAG_NO_LIST -- so that the stamps survive, and the route is forced to the one pass).
usage: python tools/decode_tile_timeline.py [sparse|dense|u9 ...]"""
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
for kind in sys.argv[1:] or ["sparse"]:
    listed = kind.startswith("u")
    if listed:
        os.environ["WAH_DIAG_NO_LIST"] = "1"
    else:
        os.environ.pop("WAH_DIAG_NO_LIST", None)
    d = (wah.gen_uniform_device(n, 1337, 2.0 ** -int(kind[1:])) if listed else
         {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5)}[kind]())
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    stream = comp.result().clone()
    del comp
    dec = wah.DeviceDecompressor(stream.numel(), n + 1, one_pass=True)
    dec.run(stream)
    dec.status()
    dec.run(stream)
    dec.status()
    batch = 1 if os.environ.get("WAH_DT_BATCH") == "1" else 2
    n_tiles = (stream.numel() + 8192 * batch - 1) // (8192 * batch)  # workgroups
    t = dec.out[: n_tiles * 32].view(torch.int64).cpu().numpy().reshape(n_tiles, 16)
    # (the stamps lie where the first workgroups' output goes: rows that were written over afterwards are dropped)
    life = t[:, 6] - t[:, 0]
    keep = (life > 0) & (life < 100_000_000) & (t[:, 1] >= t[:, 0]) & (t[:, 6] >= t[:, 5])
    print(f"({int(keep.sum())} of {n_tiles} rows kept)")
    t = t[keep]
    start, staged, pub, flags, base, bar3, end, bidx = (t[:, i] for i in range(8))
    rows = np.nonzero(keep)[0] if False else None
    loop0, setup, seg1, seg2, loop_end, bar4, loop1, nseg = (t[:, i] for i in range(8, 16))
    t0 = start.min()
    us = lambda x: x / 100.0
    q = lambda x: f"{us(np.median(x)):.2f} (p10 {us(np.percentile(x, 10)):.2f}, p90 {us(np.percentile(x, 90)):.2f})" if len(x) else "--"
    print(f"--- {kind}: {n_tiles} workgroups of {batch} x 8192 words, span {us(end.max() - t0):.1f} us")
    print(f"   start -> all tiles counted, the last one staged {q(staged - start)}")
    print(f"   -> barrier 1 (publish)                          {q(pub - staged)}")
    print(f"   -> the first tile's start flags, barrier 2      {q(flags - pub)}")
    print(f"   -> base known (wave 0)                          {q(base - flags)}")
    print(f"   -> barrier 3                                    {q(bar3 - base)}")
    print(f"   -> all tiles expanded, end                      {q(end - bar3)}")
    two = seg2 > 0
    if listed:
        two = np.zeros_like(two)
        loop0 = setup = seg1 = loop_end = bar4 = bar3
        loop1 = np.zeros_like(loop1)
    print(f"   wave 0, the batch's last tile: barrier 3 -> its segment loop   {q(loop0 - bar3)}")
    print(f"     first segment: flags, words in front (setup)                 {q(setup - loop0)}")
    print(f"     first segment: 16 steps issued                               {q(seg1 - setup)}")
    if two.any():
        print(f"     second segment (whole)                                       {q((seg2 - seg1)[two])}   ({int(two.sum())} rows)")
    print(f"     its segments done ({np.median(nseg):.0f} of them) -> the other waves' too (barrier)  {q(bar4 - loop_end)}")
    print(f"     whole segment loop of the tile                               {q(loop_end - loop0)}")
    print(f"     -> the next tile's segment loop (image, flags, counts)       {q((loop1 - bar4)[loop1 > 0])}")
    print(f"     the next tile's loop -> end                                  {q((end - loop1)[loop1 > 0])}")
    tick = np.arange(n_tiles)[keep]
    dd = bidx - tick
    print(f"   ticket == blockIdx for {100.0 * np.mean(dd == 0):.1f} % of the workgroups; |difference| median {np.median(np.abs(dd)):.0f}, p90 {np.percentile(np.abs(dd), 90):.0f}, max {np.abs(dd).max():.0f}")
    print(f"   life of the workgroup                           {q(end - start)}")
    s = us(start - t0)
    e = us(end - t0)
    for x in np.arange(0, e.max() + 50, 50.0):
        print(f"   {x:6.1f} us: in flight {int(np.sum((s <= x) & (x < e))):4d}  started {int(np.sum(s <= x)):6d}")
    del dec, d, stream
