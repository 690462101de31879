import importlib, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
n = 992 * 3000 + 5
d = wah.gen_uniform_device(n, 1, 0.01)
comp = wah.DeviceCompressor(n)
comp.run(d); torch.cuda.synchronize(); comp.status(); c0 = int(comp.count.item()); print("eager count", c0)
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        comp.run(d)
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    try:
        comp.status(); print("replay", i, "count", int(comp.count.item()))
    except Exception as e:
        print("replay", i, "error", e)
    ctrl = comp.workspace[:1024].view(torch.int32).cpu().numpy()
    print("   ctrl start", ctrl[0], "error", ctrl[160], "census", ctrl[161])
