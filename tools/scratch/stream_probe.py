import importlib, os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
wah = importlib.import_module("gpu-wah_amd")
n = 992 * 3000 + 5
src = wah.gen_clustered_device(n, 2)
comp = wah.DeviceCompressor(n)
comp.out.fill_(-2147483648)
comp.run(src); comp.status(); c = int(comp.count.item())
for pad in (c, c + 1):
    res = {}
    for route in ("1", "1000000000000"):
        os.environ["WAH_STREAM_MIN_TILES"] = route
        dec = wah.DeviceDecompressor(comp.capacity, n + 1)
        dec.out.fill_(0x5A5A5A5A)
        dec.run(comp.out, pad)
        try:
            dec.status()
        except Exception as e:
            print("pad", pad, "route", route, "error", e); continue
        res[route] = (dec.info.cpu().tolist(), dec.out[:n].clone())
        ctrl = dec.workspace[:1024].view(torch.int32).cpu().numpy()
        print("      route", route, "ctrl[170..174]", ctrl[170:174].tolist())
    ok = {r: bool(torch.equal(v[1], src)) for r, v in res.items()}
    print("pad", pad, "of", comp.capacity, "count", c, {r: v[0] for r, v in res.items()}, "equal to source:", ok)
    for r, v in res.items():
        if not ok[r]:
            bad = torch.nonzero(v[1] != src).flatten()
            print("   route", r, "first bad words", bad[:5].tolist(), "of", bad.numel(), "segment", int(bad[0]) // 992, "last bad", int(bad[-1]))
            print("   got ", [hex(x & 0xFFFFFFFF) for x in v[1][bad[:6]].tolist()], "want", [hex(x & 0xFFFFFFFF) for x in src[bad[:6]].tolist()])
