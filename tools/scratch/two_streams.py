"""Two compress launches in flight on two streams of one device: do the persistent kernels get in each other's way?"""
import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
import os
n = int(os.environ.get('N_WORDS', 268435200 // 4))
a = wah.gen_uniform_device(n, 1, 0.01); b = wah.gen_uniform_device(n, 2, 0.01)
ca, cb = wah.DeviceCompressor(n), wah.DeviceCompressor(n)
ca.run(a); cb.run(b); torch.cuda.synchronize(); ca.status(); cb.status()
want_a, want_b = int(ca.count.item()), int(cb.count.item())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t0 = time.time()
ITERS = int(os.environ.get('ITERS', 20))
for it in range(ITERS):
    with torch.cuda.stream(s1):
        ca.run(a, stream=s1)
    with torch.cuda.stream(s2):
        cb.run(b, stream=s2)
    torch.cuda.synchronize()
    try:
        ca.status(); cb.status()
    except Exception as e:
        print("iteration", it, "error:", e, f"after {time.time() - t0:.2f} s"); break
    assert int(ca.count.item()) == want_a and int(cb.count.item()) == want_b
else:
    print(f"{ITERS} concurrent pairs of {n} words ok in {time.time() - t0:.3f} s")
