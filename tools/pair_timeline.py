#!/usr/bin/env python3
"""Time line of compress_pair_kernel (DIAGNOSTIC build, WAH_TUNE=78): per tile, when it started, had its first pair's
words in registers, finished pass 1, published, finished pass 2, knew its offset (s_memrealtime, 100 MHz); tiles in
flight over time.  usage: python tools/pair_timeline.py [sparse|clustered|dense ...]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "gpu-wah_amd", "libwah_hip_diag.so")
os.environ["WAH_TUNE"] = "78"
import numpy as np  # noqa: E402
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
n = 268435200


def wall_clock_khz():
    """The rate of s_memrealtime as the runtime states it (hipDeviceAttributeWallClockRate = 10017)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    v = ctypes.c_int(0)
    rc = hip.hipDeviceGetAttribute(ctypes.byref(v), 10017, 0)
    return v.value if rc == 0 else 0


khz = wall_clock_khz() or 100000
print(f"wall clock rate: {khz} kHz")
tile_segs = 16 * int(os.environ.get("WAH_WAVE_PAIRS", "3"))
for kind in sys.argv[1:] or ["sparse"]:
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    comp.status()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    comp.run(d)
    ev[1].record()
    comp.status()
    print(f"--- {kind}: the stamped launch between two events on its stream: {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us")
    n_tiles = int(os.environ.get("N_TILES", "0")) or (270600 + tile_segs - 1) // tile_segs
    cap = comp.capacity
    # 8 x u64 per tile at the END of the output buffer, the last tile's first (wah_compress_pair.inc)
    t = comp.out[: cap].view(torch.int64)[(cap // 2) - 8 * n_tiles: cap // 2].cpu().numpy().reshape(n_tiles, 8)[::-1].astype(np.int64)
    keep = (t[:, 0] > 0) & (t[:, 7] >= t[:, 0])
    print(f"({int(keep.sum())} of {n_tiles} rows kept)")
    t = t[keep]
    start, pub, loaded, done, p1, p2, bar2, end = (t[:, i] for i in range(8))
    t0 = start.min()
    us = lambda x: x * 1e3 / khz
    q = lambda x: f"{us(x.mean()):.2f} (p10 {us(np.percentile(x, 10)):.2f}, p90 {us(np.percentile(x, 90)):.2f})"
    print(f"--- {kind}: {n_tiles} tiles, span {us(end.max() - t0):.1f} us")
    print(f"   start -> first pair in registers   {q(loaded - start)}")
    print(f"   -> wave 0's pass 1 done            {q(p1 - loaded)}")
    print(f"   -> barrier 1 passed (publish)      {q(pub - p1)}")
    print(f"   -> pass 2 + parking done           {q(p2 - pub)}")
    print(f"   -> offset known                    {q(done - p2)}")
    print(f"   -> barrier 2 passed                {q(bar2 - done)}")
    print(f"   -> every wave's stores issued      {q(end - bar2)}")
    print(f"   tile life                          {q(end - start)}")
    s = us(start - t0)
    e = us(end - t0)
    order = np.argsort(s)
    for x in np.arange(0, e.max() + 5, 5.0):
        live = (s <= x) & (x < e)
        print(f"   {x:6.1f} us: in flight {int(np.sum(live)):4d}  started {int(np.sum(s <= x)):5d}  ended {int(np.sum(e <= x)):5d}")
    del comp, d
