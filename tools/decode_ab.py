"""general decode (wah_decompress_device) on the three 1 GiB bench bitmaps: ms per decode, bit-exact check against the input.
DECODE_ROUTE=two / nowait selects the sums + expand launches (WAH_TWO_LAUNCHES / WAH_NO_WAIT); the route the library reports
is printed."""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("WAH_"))
n = 268435200
for kind in sys.argv[1:] or ["sparse", "clustered", "dense"]:
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    stream = comp.result().clone()
    del comp
    kw = {"two": {"two_launches": True}, "nowait": {"no_wait": True}}.get(os.environ.get("DECODE_ROUTE", ""), {})
    dec = wah.DeviceDecompressor(stream.numel(), n + 1, **kw)
    dec.run(stream)
    back = dec.result()
    ok = bool(torch.equal(back[:n], d)) and back.numel() in (n, n + 1)
    for _ in range(3): dec.run(stream)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): dec.run(stream)
    ev[1].record(); torch.cuda.synchronize()
    dec.status()
    ms = ev[0].elapsed_time(ev[1]) / 20
    c = stream.numel()
    print(f"[{tag}] {kind:9s} ({dec.route}): {ms:.4f} ms  roofline {(4*c+4*n)/ms/1e6/8000:.3f}  {'bit-exact' if ok else 'MISMATCH'}", flush=True)
    del dec, d, stream, back
