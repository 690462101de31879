"""compress-only time of the unsegmented encoder mode against the segmented one, 1 GiB"""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
n = 268435200
for kind in ("sparse", "clustered", "dense", "zeros"):
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337), "zeros": lambda: torch.zeros(n, dtype=torch.int32, device="cuda")}[kind]()
    res = []
    for unseg in (False, True):
        comp = wah.DeviceCompressor(n, unsegmented=unseg)
        for _ in range(3): comp.run(d)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10): comp.run(d)
        ev[1].record(); torch.cuda.synchronize()
        comp.status()
        res.append((ev[0].elapsed_time(ev[1]) / 10, int(comp.count.item())))
        del comp
    print(f"{kind}: segmented {res[0][0]:.3f} ms C={res[0][1]}, unsegmented {res[1][0]:.3f} ms C={res[1][1]}")
    del d
