#!/usr/bin/env python3
"""Times the padded-probe variants (tools/experiments/padded_probe.diff, WAH_PROBE_VARIANT) on the three 1 GiB bitmaps."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
wah = importlib.import_module("gpu-wah_amd")
n = 992 * 1024 * 264
segs = n // 992
padded = torch.empty(segs * 1024, dtype=torch.int32, device="cuda")
counts = torch.zeros(segs, dtype=torch.int32, device="cuda")
libs = {}
for v in sys.argv[1:]:
    l = ctypes.CDLL(os.path.join(ROOT, f"tools/scratch/libwah_probe{v}.so"))
    l.wah_probe_compress_padded.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    libs[v] = l
for c in range(3):
    spec = wah.columns.column_spec(c, n, seed=1337)
    col = wah.columns.make_column(wah, spec, "cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    line = f"{spec.kind:9s}:"
    for v, l in libs.items():
        fn = lambda: l.wah_probe_compress_padded(col.data_ptr(), n, padded.data_ptr(), counts.data_ptr(), s)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        line += f"  v{v} {ev[0].elapsed_time(ev[1]) / 10:.4f} ms"
    print(line, flush=True)
    del col
