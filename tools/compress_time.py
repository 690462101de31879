"""Isolated compress launches on the three 1 GiB bench bitmaps: ms per launch, roofline fraction, C and a checksum of the
stream (to compare builds / kernel variants selected by environment variables across processes).
usage: python tools/compress_time.py [size_MiB ...]   (default 1024)"""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
sizes = [int(x) for x in sys.argv[1:]] or [1024]
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("WAH_"))
for mib in sizes:
    n = mib * 1024 * 1024 // 4 // 992 * 992
    def half_dense():  # every other segment all zeros, the others p = 0.5: lanes 32-63 of every pair hold 32 words each
        t = wah.gen_uniform_device(n, 1337, 0.5)
        t.view(-1, 992)[::2] = 0
        return t
    def periodic(period):  # groups: literal, then period - 1 zero groups, ... -> every lane holds the same number of words
        import numpy as np
        bits = np.zeros(31 * period * 32, dtype=np.uint8).reshape(32, period, 31)
        bits[:, 0, ::2] = 1
        w = np.packbits(bits.reshape(-1), bitorder="little").view(np.uint32)
        reps_ = (n + w.size - 1) // w.size
        return torch.from_numpy(np.tile(w, reps_)[:n].copy()).view(torch.int32).cuda()
    kinds = os.environ.get("KINDS", "sparse clustered dense").split()
    for kind in kinds:
        d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
             "clustered": lambda: wah.gen_clustered_device(n, 1337), "p4": lambda: wah.gen_uniform_device(n, 1337, 0.25),
             "p8": lambda: wah.gen_uniform_device(n, 1337, 0.125), "p16": lambda: wah.gen_uniform_device(n, 1337, 0.0625),
             "p32": lambda: wah.gen_uniform_device(n, 1337, 1 / 32), "half_dense": half_dense, "periodic2": lambda: periodic(2), "periodic4": lambda: periodic(4)}[kind]()
        comp = wah.DeviceCompressor(n, indexed=True)
        for _ in range(3): comp.run(d)
        torch.cuda.synchronize()
        reps = 20 if mib >= 512 else 100
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps): comp.run(d)
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        c = int(comp.count.item())
        out = comp.out[:c].to(torch.int64)
        idx = torch.arange(c, device=out.device, dtype=torch.int64)
        chk = int(((out * ((idx % 1000003) + 1)).sum()).item()) & 0xFFFFFFFFFFFF
        ochk = int((comp.seg_offsets[: n // 992 + 1] * (torch.arange(n // 992 + 1, device=out.device) % 1009 + 1)).sum().item()) & 0xFFFFFFFFFFFF
        print(f"[{tag}] {mib} MiB {kind:9s}: {ms:.4f} ms  roofline {(4*n+4*c)/ms/1e6/8000:.3f}  C {c}  chk {chk:012x} idx {ochk:012x} status {comp.status()}", flush=True)
        del comp, d, out, idx
