#!/usr/bin/env python3
"""Padded-probe variant 6: compress work with KNOWN offsets (no scan): the upper bound for any placement scheme."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
wah = importlib.import_module("gpu-wah_amd")
libs = {}
for v in (sys.argv[1:] or ["6"]):  # 6: {word, position} compaction + finalize; 7: final words in LDS, plain copy out
    l = ctypes.CDLL(os.path.join(ROOT, f"tools/scratch/libwah_probe{v}.so"))
    l.wah_probe_compress_padded.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    libs[v] = l
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
    cases = [("compress_kernel", lambda: comp.run(col))]
    for v, l in libs.items():
        cases.append((f"known offsets v{v}", (lambda l: lambda: l.wah_probe_compress_padded(col.data_ptr(), n, out.data_ptr(), offs.data_ptr(), s))(l)))
    for name, fn in cases:
        if name != "compress_kernel":
            out.zero_()
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
        if name != "compress_kernel":
            assert torch.equal(out[: stream.numel()], stream), name + ": stream differs"
    print(f"{spec.kind:9s}: " + ", ".join(f"{k} {v:.4f} ms" for k, v in res.items()), flush=True)
    del col, comp, stream, offs
