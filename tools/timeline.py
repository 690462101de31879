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
W = int(os.environ.get('WAH_WORKERS', '15'))
T = (T + W - 1) // W
tl = tl[: 4 * T].reshape(T, 4).astype(np.int64)
t0 = tl[:, 0].min()
start, cls, res = [(tl[:, i] - t0) / 100.0 for i in range(3)]
wg = tl[:, 3] >> 32
it = tl[:, 3] & 0xFFFFFFFF
print(f"tiles {T}, kernel span {res.max():.1f} us")
print(f"start->publish: mean {np.mean(cls - start):.2f} us, p50 {np.median(cls - start):.2f}, p99 {np.percentile(cls - start, 99):.2f}")
print(f"publish->resolved: mean {np.mean(res - cls):.2f} us, p50 {np.median(res - cls):.2f}, p99 {np.percentile(res - cls, 99):.2f}")
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

G = int(wg.max()) + 1
print("grid", G)
# per generation: spread of publish times, and resolve overhead beyond the latest predecessor publish
for g in (1, 5, 10, 20, 40):
    m = np.where(it == g)[0]
    if len(m) < G:
        continue
    pub = cls[m]
    r = res[m]
    runmax = np.maximum.accumulate(pub)  # latest publish among slots <= i
    print(f"  gen {g}: publish min {pub.min():.1f} p50 {np.median(pub):.1f} p90 {np.percentile(pub,90):.1f} max {pub.max():.1f} | "
          f"resolve - latest predecessor publish: p50 {np.median(r - runmax):.2f} us p90 {np.percentile(r - runmax, 90):.2f} max {(r - runmax).max():.2f}")
# is slowness systematic per workgroup?
dur = cls - start
per_wg = np.array([dur[wg == w].mean() for w in range(G)])
print(f"per-WG mean start->publish: min {per_wg.min():.2f} p10 {np.percentile(per_wg,10):.2f} p50 {np.median(per_wg):.2f} p90 {np.percentile(per_wg,90):.2f} max {per_wg.max():.2f} us")
per_wg_sd = np.array([dur[wg == w].std() for w in range(G)])
print(f"within-WG std of start->publish: median {np.median(per_wg_sd):.2f} us")
lag = np.array([(cls[wg == w] - np.array([cls[it == g].min() for g in it[wg == w]])).mean() for w in range(G)])
print(f"per-WG mean lag behind the first publisher of its generation: min {lag.min():.2f} p50 {np.median(lag):.2f} p90 {np.percentile(lag,90):.2f} max {lag.max():.2f}")
slot = np.array([np.where(wg == w)[0].min() for w in range(G)])
order = np.argsort(lag)
print("slowest WGs (blockIdx, slot, lag):", [(int(w), int(slot[w]), round(float(lag[w]), 2)) for w in order[-8:]])
print("fastest WGs (blockIdx, slot, lag):", [(int(w), int(slot[w]), round(float(lag[w]), 2)) for w in order[:8]])
print("lag by blockIdx % 8 (XCD group):", [round(float(lag[np.arange(G) % 8 == x].mean()), 2) for x in range(8)])

# per-workgroup phase totals (thread 0) and XCC id
raw = comp.seg_offsets.cpu().numpy().astype(np.int64)
pw = raw[4 * T: 4 * T + 10 * G].reshape(G, 10)
names = ["stage", "read+masks", "deliver", "wait offset", "emit", "compact+finalize"]
xcc = pw[:, 8] & 0xF
tiles_wg = np.maximum(pw[:, 7], 1)
print("per-WG cycles/tile by phase: (min / p50 / max over workgroups)")
for i, nm in enumerate(names):
    v = pw[:, i] / tiles_wg
    print(f"   {nm:18s} {v.min():8.0f} {np.median(v):8.0f} {v.max():8.0f}")
own = (pw[:, 0] + pw[:, 1] + pw[:, 2] + pw[:, 4] + pw[:, 5]) / tiles_wg
print("own work cycles/tile (no waiting): min %.0f p50 %.0f p90 %.0f max %.0f" % (own.min(), np.median(own), np.percentile(own, 90), own.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"   XCC {x}: {m.sum():3d} WGs, own work {own[m].mean():8.0f} cyc/tile, stage {(pw[m,0]/tiles_wg[m]).mean():6.0f}, masks {(pw[m,1]/tiles_wg[m]).mean():6.0f}, compact {(pw[m,5]/tiles_wg[m]).mean():6.0f}, wait {(pw[m,3]/tiles_wg[m]).mean():6.0f}")

# event table for a few slots: publish P, needed-by-resolve M = latest publish it depends on, resolved R
pubg = {}
for g in range(int(it.max()) + 1):
    m = np.where(it == g)[0]
    arr = np.full(G, np.nan)
    arr[slot_of := (m - g * G)] = cls[m]
    pubg[g] = arr
for sl in (0, 1, 64, 128, 200, 255):
    row = []
    for g in range(20, 26):
        t = g * G + sl
        if t >= T:
            continue
        cur = pubg[g][:sl]
        prv = pubg[g - 1][sl + 1:]
        M = np.nanmax(np.concatenate([cur, prv])) if (len(cur) + len(prv)) else float('nan')
        row.append(f"g{g}: P {cls[t]:7.1f} M {M:7.1f} R {res[t]:7.1f} (R-P {res[t]-cls[t]:4.1f}, R-M {res[t]-M:4.1f})")
    print(f"slot {sl:3d} " + " | ".join(row))

g = 20
P = pubg[g]
Rg = {gg: np.array([res[gg * G + sl] if gg * G + sl < T else np.nan for sl in range(G)]) for gg in (g - 2, g - 1, g)}
order = np.argsort(P)
print("gen 20 publish order (slot:P | R(g-2) R(g-1)):")
print(" earliest:", [f"{int(sl)}:{P[sl]:.1f}|{Rg[g-2][sl]:.1f} {Rg[g-1][sl]:.1f}" for sl in order[:10]])
print(" latest  :", [f"{int(sl)}:{P[sl]:.1f}|{Rg[g-2][sl]:.1f} {Rg[g-1][sl]:.1f}" for sl in order[-10:]])
print(" P - R(g-2) over slots: min %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.nanpercentile(P - Rg[g-2], [0, 50, 90, 100])))
print(" by slot decile, mean P:", [round(float(np.nanmean(P[d * G // 10:(d + 1) * G // 10])), 1) for d in range(10)])
print(" by slot decile, mean R(g-2):", [round(float(np.nanmean(Rg[g-2][d * G // 10:(d + 1) * G // 10])), 1) for d in range(10)])

pv = raw[4 * T + 10 * G: 4 * T + 10 * G + 64 * 16 * 8].reshape(64, 16, 8)
W = int(os.environ.get('WAH_WORKERS', '15'))
print("per-wave cycles/tile (mean over 64 WGs): wave: stage masks deliver wait emit compact | own")
for w in range(W):
    v = pv[:, w, :].astype(float)
    t = np.maximum(v[:, 7], 1)[:, None]
    m = (v / t).mean(axis=0)
    print(f"   wave {w:2d}: {m[0]:6.0f} {m[1]:6.0f} {m[2]:6.0f} {m[3]:6.0f} {m[4]:6.0f} {m[5]:6.0f} | {m[0]+m[1]+m[2]+m[4]+m[5]:6.0f}")
