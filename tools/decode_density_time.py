"""The general decoder by bit density (uniform bitmaps, one bit in 2^i, 992 MiB as the reference's report sweep): the one-pass route
(decode_tile_kernel + the list's launch, WAH_ONE_PASS) against the two launches (WAH_TWO_LAUNCHES), ms per decode, bit-exact check.  Between about
7 and 30 groups per word every tile goes onto the list and is expanded by work items; this is where the two routes differ most.
usage: python tools/decode_density_time.py [i ...]   (default 7..13)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wah = importlib.import_module("gpu-wah_amd")
n = 256 * 1024 * 992
for i in [int(x) for x in sys.argv[1:]] or list(range(7, 14)):
    d = wah.gen_uniform_device(n, 1337, 2.0 ** -i)
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    stream = comp.result().clone()
    del comp
    row = []
    for kw in ({"one_pass": True}, {"two_launches": True}):
        dec = wah.DeviceDecompressor(stream.numel(), n + 1, **kw)
        dec.run(stream)
        ok = bool(torch.equal(dec.result()[:n], d))
        for _ in range(3): dec.run(stream)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10): dec.run(stream)
        ev[1].record(); torch.cuda.synchronize()
        dec.status()
        row.append(f"{dec.route}: {ev[0].elapsed_time(ev[1]) / 10:.4f} ms{'' if ok else ' MISMATCH'}")
        del dec
    dec = wah.DeviceDecompressor(stream.numel(), n + 1)
    dec.run(stream)
    print(f"density 2^-{i}: C/N {stream.numel() / n:.4f} ({32 / 31 * n / stream.numel():.1f} groups per word)  " + "   ".join(row) + f"   (default at this capacity: {dec.route})", flush=True)
    del dec
    del d, stream
