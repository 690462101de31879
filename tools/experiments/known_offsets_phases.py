#!/usr/bin/env python3
"""Worker phase shares (diagnostic stamps) of compress_kernel with the scan and with table offsets."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "tools/scratch/libwah_known_diag.so")
import torch
wah = importlib.import_module("gpu-wah_amd")
names = ["wait loads + stage", "prefetch + read + masks", "deliver count (+publish)", "wait for previous offset", "emit previous tile", "compact + finalize", "-"]
n = 268435200
col = wah.gen_uniform_device(n, 1337, 0.01)
# the table comes from the PRODUCT library's indexed compressor (the diagnostic build uses seg_offsets for its stamps)
import ctypes
prod = importlib.import_module("gpu-wah_amd")
comp0 = wah.DeviceCompressor(n)
comp0.run(col); comp0.status()
stream = comp0.result().clone()
# offsets of the segments: decode-free reconstruction through the validator is overkill; take them from a plain scan of
# per-segment counts computed by compressing each 15-segment tile? -> simpler: product build in a subprocess
import subprocess, tempfile
f = tempfile.mktemp(suffix=".pt")
code = (f"import importlib,torch; w=importlib.import_module('gpu-wah_amd'); d=w.gen_uniform_device({n},1337,0.01); "
        f"c=w.DeviceCompressor({n},indexed=True); c.run(d); c.status(); torch.save(c.seg_offsets.cpu(), '{f}')")
env = dict(os.environ); env.pop("WAH_LIB_PATH")
subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, check=True)
offs = torch.load(f).cuda()
comp = wah.DeviceCompressor(n)
for name in ("scan", "table"):
    if name == "table":
        os.environ["WAH_EXP_KNOWN_PTR"] = str(offs.data_ptr())
    else:
        os.environ.pop("WAH_EXP_KNOWN_PTR", None)
    comp.run(col); comp.status()
    comp.run(col); comp.status()
    assert torch.equal(comp.result(), stream), name
    acc = comp.workspace[768:768 + 256].view(torch.int64).cpu().tolist()
    tiles = max(acc[7], 1)
    total = sum(acc[:7])
    print(f"--- {name}: {tiles} tiles, {total / tiles:.0f} cycles/tile (worker 0 of each workgroup)")
    for nm, v in zip(names[:6], acc[:6]):
        print(f"   {nm:26s} {v / tiles:9.0f} cyc/tile  {100.0 * v / total:5.1f} %")
    print(f"   scan wave per tile: wait for counts {acc[8] / tiles:.0f} cyc, resolve {acc[9] / tiles:.0f} cyc")
os.environ.pop("WAH_EXP_KNOWN_PTR", None)
