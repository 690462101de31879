"""compress-only device time by bitmap size and segments per wave (WAH_WAVE_SEGS forced by the caller)"""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
print("WAH_WAVE_SEGS", os.environ.get("WAH_WAVE_SEGS", "auto"))
for mib in (4, 8, 16, 32, 64, 128, 256, 512):
    n = mib * 1024 * 1024 // 4 // 992 * 992
    d = wah.gen_uniform_device(n, 1337, 0.01)
    comp = wah.DeviceCompressor(n)
    for _ in range(3): comp.run(d)
    torch.cuda.synchronize()
    reps = 50
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): comp.run(d)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    c = int(comp.count.item())
    print(f"{mib:4d} MiB: {ms*1e3:8.1f} us/launch  input {4*n/ms/1e6:8.1f} GB/s  roofline {(4*n+4*c)/ms/1e6/8000:.3f}")
    del comp, d
