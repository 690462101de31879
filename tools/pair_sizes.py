"""compress-only device time by bitmap size for the pair kernel's tile shapes (WAH_WAVE_PAIRS forced by the caller).
SIZES=1024,2048,4096 (MiB) in the environment: other sizes than the default 1 .. 512 MiB (a bitmap of up to 256 MiB lies in the
memory-side cache between the back-to-back launches: their fractions are not those of a cold launch)."""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("WAH_"))
kinds = sys.argv[1:] or ["sparse"]
for kind in kinds:
    row = []
    for mib in [int(x) for x in os.environ.get("SIZES", "1,4,8,16,32,64,128,256,512").split(",")]:
        n = mib * 1024 * 1024 // 4 // 992 * 992
        d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
             "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
        comp = wah.DeviceCompressor(n)
        for _ in range(5): comp.run(d)
        torch.cuda.synchronize()
        reps = 100 if mib <= 512 else 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps): comp.run(d)
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        c = int(comp.count.item())
        comp.status()
        row.append(f"{mib}M {ms*1e3:.1f}us ({(4*n+4*c)/ms/1e6/8000:.2f})")
        del comp, d
    print(f"[{tag}] {kind}: " + "  ".join(row), flush=True)
