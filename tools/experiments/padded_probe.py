#!/usr/bin/env python3
"""Padded probe (tools/experiments/README.md): compress work with one short-lived wavefront per segment, no offset scan."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
wah = importlib.import_module("gpu-wah_amd")
probe = ctypes.CDLL(os.path.join(ROOT, "tools/scratch/libwah_probe0.so"))
probe.wah_probe_compress_padded.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
n = 992 * 1024 * 264
segs = n // 992
padded = torch.empty(segs * 1024, dtype=torch.int32, device="cuda")
counts = torch.zeros(segs, dtype=torch.int32, device="cuda")
for c in range(3):
    spec = wah.columns.column_spec(c, n, seed=1337)
    col = wah.columns.make_column(wah, spec, "cuda:0")
    comp = wah.DeviceCompressor(n, indexed=True)
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, fn in (("compress_kernel", lambda: comp.run(col)),
                     ("padded probe", lambda: probe.wah_probe_compress_padded(col.data_ptr(), n, padded.data_ptr(), counts.data_ptr(), s))):
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
    stream = comp.result()
    offs = comp.seg_offsets
    assert torch.equal(counts.long(), offs[1:] - offs[:-1]), "counts differ"
    # the padded slots hold exactly the stream's words
    idx = torch.arange(stream.numel(), device="cuda")
    seg_of = torch.repeat_interleave(torch.arange(segs, device="cuda"), counts.long())
    assert torch.equal(padded[seg_of * 1024 + (idx - offs[seg_of])], stream), "words differ"
    print(f"{spec.kind:9s}: compress_kernel {res['compress_kernel']:.4f} ms, padded probe {res['padded probe']:.4f} ms", flush=True)
    del col, comp, stream, offs, idx, seg_of
