#!/usr/bin/env python3
"""Phase times of the compress tile kernel from the DIAGNOSTIC build (make -C gpu-wah_amd diag): averages over the
tiles of one launch, from s_memrealtime stamps of wave 0 (100 MHz).  Shares only -- never quote this build's run time."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "gpu-wah_amd", "libwah_hip_diag.so")
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
names = ["loads -> staged", "classify", "barrier 1", "scan", "polls", "tiles", "lifetime"]
n = 268435200
for kind in sys.argv[1:] or ["sparse", "dense", "clustered"]:
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    comp.status()
    comp.workspace[768:1024].zero_()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    comp.run(d)
    ev[1].record()
    comp.status()
    acc = comp.workspace[768:1024].view(torch.int64).cpu().tolist()
    tiles = max(acc[5], 1)
    print(f"--- {kind}: {tiles} tiles, launch {ev[0].elapsed_time(ev[1]):.3f} ms (diag build), WAH_TUNE={os.environ.get('WAH_TUNE', '0')}")
    for i in (0, 1, 2, 3, 6):
        print(f"   {names[i]:18s} {acc[i] / tiles / 100.0:7.2f} us/tile")
    print(f"   re-polls per tile  {acc[4] / tiles:7.2f}   tiles not on XCD blockIdx%8: {acc[7]}")
    print("   by XCD: scan us   " + " ".join(f"{acc[8 + x] / max(acc[24 + x], 1) / 100.0:6.2f}" for x in range(8)))
    print("   by XCD: re-polls  " + " ".join(f"{acc[16 + x] / max(acc[24 + x], 1):6.2f}" for x in range(8)))
    print("   by XCD: tiles     " + " ".join(f"{acc[24 + x]:6d}" for x in range(8)))
    del comp, d
