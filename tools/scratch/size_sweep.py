"""Device-API time against bitmap size (resident input, reusable workspace): the fixed cost of a call."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
for mb in (1, 4, 16, 64, 256, 1024):
    n = mb * 1024 * 1024 // 4 // 992 * 992
    d = wah.gen_uniform_device(n, 7, 0.01)
    comp = wah.DeviceCompressor(n)
    comp.run(d); c = comp.result().clone()
    dec = wah.DeviceDecompressor(c.numel(), n + 1)
    for _ in range(3):
        comp.run(d); dec.run(c)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps = 20
    ev[0].record()
    for _ in range(reps): comp.run(d)
    ev[1].record()
    for _ in range(reps): dec.run(c)
    ev[2].record()
    torch.cuda.synchronize()
    print(f"{mb:5d} MiB: compress {ev[0].elapsed_time(ev[1]) / reps * 1e3:8.1f} us   decompress {ev[1].elapsed_time(ev[2]) / reps * 1e3:8.1f} us", flush=True)
