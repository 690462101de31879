#!/usr/bin/env python3
"""compress_kernel with offsets from a table (WAH_EXP_KNOWN build): the persistent kernel minus the cost of the scan."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["WAH_LIB_PATH"] = os.path.join(ROOT, "tools/scratch/libwah_known.so")
import torch
wah = importlib.import_module("gpu-wah_amd")
n = 992 * 1024 * 264
for c in range(3):
    spec = wah.columns.column_spec(c, n, seed=1337)
    col = wah.columns.make_column(wah, spec, "cuda:0")
    comp = wah.DeviceCompressor(n, indexed=True)
    os.environ.pop("WAH_EXP_KNOWN_PTR", None)
    comp.run(col)
    stream = comp.result().clone()
    offs = comp.seg_offsets.clone()
    res = {}
    for name in ("scan", "table"):
        if name == "table":
            os.environ["WAH_EXP_KNOWN_PTR"] = str(offs.data_ptr())
        else:
            os.environ.pop("WAH_EXP_KNOWN_PTR", None)
        for _ in range(3):
            comp.run(col)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            comp.run(col)
        ev[1].record()
        torch.cuda.synchronize()
        res[name] = ev[0].elapsed_time(ev[1]) / 10
        assert torch.equal(comp.result(), stream), name
    os.environ.pop("WAH_EXP_KNOWN_PTR", None)
    print(f"{spec.kind:9s}: " + ", ".join(f"{k} {v:.4f} ms" for k, v in res.items()), flush=True)
    del col, comp, stream, offs
