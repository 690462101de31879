"""index decode (wah_decompress_indexed_device, decode_segments_kernel) on the three 1 GiB bench bitmaps: ms per launch."""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("WAH_"))
n = 268435200
for kind in sys.argv[1:] or ["sparse", "clustered", "dense"]:
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n, indexed=True)
    comp.run(d)
    stream = comp.result()
    out = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(wah.lib().wah_decompress_segments_workspace_bytes()), dtype=torch.uint8, device="cuda")
    run = lambda: wah.decompress_segments_device(stream, comp.seg_offsets, n, out=out, workspace=ws, check=False)
    back = run()
    ok = bool(torch.equal(back[:n], d))
    for _ in range(3): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10): run()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    c = int(comp.result().numel())
    print(f"[{tag}] {kind:9s}: {ms:.4f} ms  roofline {(4*c+4*n)/ms/1e6/8000:.3f}  {'bit-exact' if ok else 'MISMATCH'}", flush=True)
    del comp, d, back
