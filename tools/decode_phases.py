#!/usr/bin/env python3
"""Phase times of decode_expand_kernel from the DIAGNOSTIC build (make -C gpu-wah_amd diag): averages over a sample of
the tiles of one launch, s_memrealtime stamps of wave 0 (100 MHz).  Shares only -- never quote this build's run time."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "gpu-wah_amd", "libwah_hip_diag.so")
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
n = 268435200
for kind in sys.argv[1:] or ["sparse", "dense"]:
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    c = comp.result().numel()
    dec = wah.DeviceDecompressor(comp.capacity, n + 1)
    dec.run(comp.out, c)
    dec.status()
    dec.workspace[896:1024].zero_()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    dec.run(comp.out, c)
    ev[1].record()
    dec.status()
    acc = dec.workspace[896:1024].view(torch.int64).cpu().tolist()
    tiles = max(acc[2], 1)
    print(f"--- {kind}: {tiles} sampled tiles, sums + expand {ev[0].elapsed_time(ev[1]):.3f} ms (diag build)")
    print(f"   tile staged + coarse prefix {acc[0] / tiles / 100.0:6.2f} us/tile")
    print(f"   its segments expanded       {acc[1] / tiles / 100.0:6.2f} us/tile")
    del comp, dec, d
