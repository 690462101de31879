#!/usr/bin/env python3
"""Probe variant 8: final-word compaction + decoupled look-back over tiles of 8 segments (real offsets, one pass)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
wah = importlib.import_module("gpu-wah_amd")
l = ctypes.CDLL(os.path.join(ROOT, "tools/scratch/libwah_probe8.so"))
l.wah_probe_compress_padded.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
n = 992 * 1024 * 264
tiles = (n // 992 + 7) // 8
out = torch.zeros(wah.max_compressed_words(n) + 1024, dtype=torch.int32, device="cuda")
desc = torch.zeros(2 * tiles + 256, dtype=torch.int32, device="cuda")
for c in range(3):
    spec = wah.columns.column_spec(c, n, seed=1337)
    col = wah.columns.make_column(wah, spec, "cuda:0")
    comp = wah.DeviceCompressor(n)
    comp.run(col)
    stream = comp.result().clone()
    s = torch.cuda.current_stream().cuda_stream

    def probe():
        desc.zero_()
        l.wah_probe_compress_padded(col.data_ptr(), n, out.data_ptr(), desc.data_ptr(), s)

    res = {}
    for name, fn in (("compress_kernel", lambda: comp.run(col)), ("look-back probe", probe)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        res[name] = ev[0].elapsed_time(ev[1]) / 10
    err = int(desc[2 * tiles + 64].item())
    total = int(desc[2 * tiles + 128: 2 * tiles + 130].view(torch.int64).item())
    assert err == 0, f"error bits {err:#x}"
    assert total == stream.numel(), (total, stream.numel())
    assert torch.equal(out[: stream.numel()], stream), "stream differs"
    print(f"{spec.kind:9s}: " + ", ".join(f"{k} {v:.4f} ms" for k, v in res.items()), flush=True)
    del col, comp, stream
