"""General decoder on the three 1 GiB bitmaps: ms per decode (back-to-back launches) and a round-trip check.
"""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
n = int(os.environ.get("N_WORDS", 268435200))
for kind in ("sparse", "clustered", "dense"):
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5), "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    c = comp.result().numel()
    dec = wah.DeviceDecompressor(comp.capacity, n + 1)
    for _ in range(3): dec.run(comp.out, c)
    dec.status()
    assert int(dec.info[0].item()) == n and bool(torch.equal(dec.out[:n], d)), kind
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10): dec.run(comp.out, c)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    print(kind, "C/N %.4f" % (c / n), "decode ms %.4f" % ms, "GB/s (4C+4N) %.0f" % ((4 * c + 4 * n) / ms / 1e6), flush=True)
    del comp, dec, d
