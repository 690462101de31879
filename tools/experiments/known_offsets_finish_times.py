#!/usr/bin/env python3
"""Per-workgroup finish times of compress_kernel (WAH_EXP_KNOWN build), with the scan and with table offsets."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "tools/scratch/libwah_known.so")
import numpy as np
import torch
wah = importlib.import_module("gpu-wah_amd")
n = 992 * 1024 * 264
col = wah.gen_uniform_device(n, 1337, 0.01)
comp = wah.DeviceCompressor(n, indexed=True)
comp.run(col)
comp.status()
offs = comp.seg_offsets.clone()
for name in ("scan", "table"):
    if name == "table":
        os.environ["WAH_EXP_KNOWN_PTR"] = str(offs.data_ptr())
    else:
        os.environ.pop("WAH_EXP_KNOWN_PTR", None)
    for rep in range(3):
        comp.run(col)
        comp.status()
    raw = comp.workspace[1024 + 4 * 20000: 1024 + 4 * 20000 + 16 * 256].view(torch.int64).cpu().numpy().reshape(256, 2)
    t = (raw[:, 0] - raw[:, 0].min()) / 100.0  # us (100 MHz)
    xcc = (raw[:, 1] >> 32) & 15
    arr = raw[:, 1] & 0xFFFFFFFF
    print(f"{name}: finish spread of worker 0 over the 256 workgroups: p0 0, p10 {np.percentile(t, 10):.1f}, p50 {np.percentile(t, 50):.1f}, p90 {np.percentile(t, 90):.1f}, max {t.max():.1f} us")
    print("   mean finish by XCD:", [f"{t[xcc == x].mean():.1f}" for x in range(8)], " workgroups per XCD:", [int((xcc == x).sum()) for x in range(8)])
    late = np.argsort(t)[-8:]
    print("   latest:", [(int(b), int(xcc[b]), int(arr[b]), round(float(t[b]), 1)) for b in late])
os.environ.pop("WAH_EXP_KNOWN_PTR", None)
