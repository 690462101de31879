"""general decode of an incompressible 1 GiB bitmap with ONE long hole (a fill of millions of groups: one deferred tile of the
one-pass decoder, shared out over the second launch in parts) -- against WAH_DECODE_TWO_PASS=1"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wah = importlib.import_module("gpu-wah_amd")
n = 268435200
d = wah.gen_uniform_device(n, 1337, 0.5)
d[992 * 50000: 992 * 150000] = 0  # 100 000 segments of zeros = 102 M groups
comp = wah.DeviceCompressor(n, unsegmented=True)  # (classic form: the hole is ONE word)
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
