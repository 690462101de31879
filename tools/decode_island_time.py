"""general decode of a mostly EMPTY 1 GiB bitmap with dense islands, in its classic (unsegmented) form: a short stream (the
two-launch route) whose tiles around the islands hold fills of millions of groups -- shared out over the list workgroups of
the expand launch."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wah = importlib.import_module("gpu-wah_amd")
n = 268435200
d = torch.zeros(n, dtype=torch.int32, device="cuda")
for k in range(4):
    lo = 992 * (30000 + 60000 * k)
    d[lo: lo + 992 * 1000] = wah.gen_uniform_device(992 * 1000, 7 + k, 0.5)  # four dense islands of 4 MB
comp = wah.DeviceCompressor(n, unsegmented=True)
comp.run(d)
stream = comp.result().clone()
dec = wah.DeviceDecompressor(stream.numel(), n + 1)
dec.run(stream)
ok = bool(torch.equal(dec.result()[:n], d))
for _ in range(3): dec.run(stream)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10): dec.run(stream)
ev[1].record(); torch.cuda.synchronize()
dec.status()
print(f"[{' '.join(k + '=' + v for k, v in os.environ.items() if k.startswith('WAH_'))}] C = {stream.numel()} words: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms  {'bit-exact' if ok else 'MISMATCH'}")
