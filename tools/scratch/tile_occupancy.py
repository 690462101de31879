import importlib, os, sys
sys.path.insert(0, "/root/repo")
os.environ["WAH_LIB_PATH"] = "/root/repo/gpu-wah_amd/libwah_hip_diag.so"
os.environ["WAH_TUNE"] = "78"
import numpy as np, torch
wah = importlib.import_module("gpu-wah_amd")
n = 268435200
d = wah.gen_uniform_device(n, 1337, 0.01)
comp = wah.DeviceCompressor(n, indexed=True)
comp.run(d); comp.status(); comp.run(d); comp.status()
n_tiles = (270600 + 39) // 40
t = comp.seg_offsets[: n_tiles * 8].cpu().numpy().reshape(n_tiles, 8).astype(np.int64)
start, bar2 = t[:, 0], t[:, 6]
t0 = start.min()
s = (start - t0) / 100.0; e = (bar2 - t0) / 100.0
print("span", e.max(), "mean life (no emission)", (e - s).mean())
for x in np.arange(0, e.max() + 5, 5.0):
    print(f"{x:6.1f} us: in flight {int(np.sum((s <= x) & (x < e))):4d}  started {int(np.sum(s <= x)):5d}")
# gap between a tile's end and the next start on the same slot: approximate by sorting ends and starts
es = np.sort(e); ss = np.sort(s)[512:]
k = min(len(es), len(ss))
print("mean (start of tile 512+i) - (i-th end):", float(np.mean(ss[:k] - es[:k])))
