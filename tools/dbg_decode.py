import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
wah = importlib.import_module("gpu-wah_amd")
for mib in (int(x) for x in sys.argv[1:]):
    n = (mib << 20) // 4 // 992 * 992
    d = wah.gen_uniform_device(n, 1337, 0.01)
    comp = wah.DeviceCompressor(n)
    comp.run(d); c = comp.result()
    dec = wah.DeviceDecompressor(c.numel(), n + 1)
    try:
        dec.run(c)
        out = dec.result()
        ok = bool(torch.equal(out[:n], d))
        print(mib, "MiB: C", c.numel(), "tiles", (c.numel() + 4095) // 4096, "ok", ok, "info", dec.info.tolist())
    except Exception as e:
        ws = dec.workspace[:1024].view(torch.int32).cpu()
        print(mib, "MiB: FAIL", e, "ctrl start", int(ws[0]), "census", int(ws[161]), "err", int(ws[160]), "info", dec.info.tolist())
