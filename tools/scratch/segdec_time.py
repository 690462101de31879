#!/usr/bin/env python3
"""Index decode (wah_decompress_segments_device) against the general decoder, 1 GiB bitmaps."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
wah = importlib.import_module("gpu-wah_amd")
n = 992 * 1024 * 264
reps = 20
for c in range(3):
    spec = wah.columns.column_spec(c, n, seed=1337)
    col = wah.columns.make_column(wah, spec, "cuda:0")
    comp = wah.DeviceCompressor(n, indexed=True)
    comp.run(col)
    stream = comp.result().clone()
    offs = comp.seg_offsets.clone()
    C = stream.numel()
    dec = wah.DeviceDecompressor(C, n + 1)
    out = torch.empty(n + 1, dtype=torch.int32, device="cuda:0")
    ws = torch.empty(int(wah.lib().wah_decompress_segments_workspace_bytes()), dtype=torch.uint8, device="cuda:0")
    res = {}
    for name, fn in (("general", lambda: dec.run(stream)),
                     ("indexed", lambda: wah.decompress_segments_device(stream, offs, n, out=out, workspace=ws, check=False))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
        res[name] = ts[len(ts) // 2]
    assert torch.equal(out[:n], col) and torch.equal(dec.result()[:n], col)
    algo = 4.0 * C + 4.0 * n
    print(f"{spec.kind:9s} C/N {C / n:.4f}: general {res['general']:.4f} ms, indexed {res['indexed']:.4f} ms "
          f"({algo / res['indexed'] / 1e6:.0f} GB/s algorithmic = {algo / res['indexed'] / 1e6 / 8000:.3f} of 8 TB/s)")
    del col, comp, stream, offs, dec, out
