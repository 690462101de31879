"""How much slower is the decoder when the stream contains a fill word of count 0 (index-map route)?"""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
n = 268435200
d = wah.gen_uniform_device(n, 1337, 0.01)
c = wah.compress_device(d)
for name, stream in (("as compressed", c), ("one empty fill appended", torch.cat([c, torch.tensor([-2147483648], dtype=torch.int32, device="cuda")]))):
    dec = wah.DeviceDecompressor(stream.numel(), n + 1)
    for _ in range(2): dec.run(stream)
    torch.cuda.synchronize(); dec.status()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): dec.run(stream)
    ev[1].record(); torch.cuda.synchronize()
    assert bool(torch.equal(dec.result()[:n], d))
    print(f"{name}: decompress {ev[0].elapsed_time(ev[1]) / 5:.3f} ms", flush=True)
