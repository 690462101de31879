import importlib, os, sys, torch, numpy as np
os.environ["WAH_STREAM_MIN_TILES"] = "1"
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
n = 992 * 3000 + 5
a = wah.gen_uniform_device(n, 1, 0.01); b = wah.gen_clustered_device(n, 2)
d_in = a.clone()
comp = wah.DeviceCompressor(n); dec = wah.DeviceDecompressor(comp.capacity, n + 1)
comp.out.fill_(-2147483648)
comp.run(d_in); dec.run(comp.out, comp.capacity); torch.cuda.synchronize(); comp.status(); dec.status()
print("eager ok:", bool(torch.equal(dec.out[:n], a)))
side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        comp.run(d_in); dec.run(comp.out, comp.capacity)
for i, src in enumerate((a, b, a, b)):
    comp.out.fill_(-2147483648); d_in.copy_(src); dec.out.fill_(0x5A5A5A5A)
    g.replay(); torch.cuda.synchronize()
    try:
        comp.status(); dec.status()
    except Exception as e:
        print("replay", i, "error", e); continue
    ok = bool(torch.equal(dec.out[:n], src))
    bad = torch.nonzero(dec.out[:n] != src).flatten()
    ctrl = dec.workspace[:1024].view(torch.int32).cpu().numpy()
    print("replay", i, "equal:", ok, "info", dec.info.cpu().tolist(), "tickets", ctrl[0], "bad words", bad.numel(), bad[:4].tolist(), bad[-2:].tolist() if bad.numel() else "")
