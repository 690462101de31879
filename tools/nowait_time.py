"""The no-wait compress route (WAH_NO_WAIT: count, scan, place) against the one-launch kernel, 1 GiB bitmaps."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
n = 268435200
for kind in ("sparse", "clustered", "dense"):
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5), "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    res = []
    for no_wait in (False, True):
        comp = wah.DeviceCompressor(n, no_wait=no_wait)
        for _ in range(3): comp.run(d)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10): comp.run(d)
        ev[1].record(); torch.cuda.synchronize()
        res.append(ev[0].elapsed_time(ev[1]) / 10)
        del comp
    print(kind, "one launch %.4f ms, no-wait route %.4f ms" % tuple(res), flush=True)
    del d
