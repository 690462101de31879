#!/usr/bin/env python3
"""Padded-probe variant 6: compress work with KNOWN offsets (no scan): the upper bound for any placement scheme."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
wah = importlib.import_module("gpu-wah_amd")
l = ctypes.CDLL(os.path.join(ROOT, "tools/scratch/libwah_probe6.so"))
l.wah_probe_compress_padded.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
n = 992 * 1024 * 264
out = torch.zeros(wah.max_compressed_words(n) + 1024, dtype=torch.int32, device="cuda")
for c in range(3):
    spec = wah.columns.column_spec(c, n, seed=1337)
    col = wah.columns.make_column(wah, spec, "cuda:0")
    comp = wah.DeviceCompressor(n, indexed=True)
    comp.run(col)
    stream = comp.result().clone()
    offs = comp.seg_offsets.clone()
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, fn in (("compress_kernel", lambda: comp.run(col)),
                     ("known offsets", lambda: l.wah_probe_compress_padded(col.data_ptr(), n, out.data_ptr(), offs.data_ptr(), s))):
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
    assert torch.equal(out[: stream.numel()], stream), "stream differs"
    print(f"{spec.kind:9s}: compress_kernel {res['compress_kernel']:.4f} ms, known offsets {res['known offsets']:.4f} ms", flush=True)
    del col, comp, stream, offs
