#!/usr/bin/env python3
"""Per-phase cycle shares of the compress kernel from the DIAGNOSTIC build (make -C gpu-wah_amd diag).
Reads the stamp totals the kernel added into the control block.  Shares only -- never quote this build's run time."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "gpu-wah_amd", "libwah_hip_diag.so")
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
names = ["wait loads + stage", "prefetch + read + masks", "deliver count (+publish)", "wait for previous offset", "emit previous tile", "compact + finalize", "-", "tiles"]
n = 268435200
for kind in sys.argv[1:] or ["sparse", "dense", "clustered"]:
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    comp.status()
    comp.run(d)
    comp.status()
    acc = comp.workspace[768:768 + 256].view(torch.int64).cpu().tolist()
    tiles = max(acc[7], 1)
    total = sum(acc[:7])
    print(f"--- {kind}: {tiles} tiles, {total / tiles:.0f} cycles/tile (thread 0 of each workgroup)")
    for nm, v in zip(names[:7], acc[:7]):
        print(f"   {nm:18s} {v / tiles:9.0f} cyc/tile  {100.0 * v / total:5.1f} %")
    big = acc[16:24]
    print("   iterations (of thread 0) in which the phase took > 4000 cycles: " + ", ".join(f"{nm.split()[0]} {100.0 * b / tiles:.1f}%" for nm, b in zip(names[:6], big[:6])))
    print(f"   scan wave per tile: wait for counts {acc[8] / tiles:.0f} cyc, resolve {acc[9] / tiles:.0f} cyc")
    del comp, d
