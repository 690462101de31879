"""time of the sums pass alone (wah_decompress_scan_device) on the sparse / dense 1 GiB streams"""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
lib = wah.lib()
n = 268435200
print(os.environ.get("WAH_LIB_PATH", "product"))
for kind in ("sparse", "dense", "clustered"):
    d = {"sparse": lambda: wah.gen_uniform_device(n, 1337, 0.01), "dense": lambda: wah.gen_uniform_device(n, 1337, 0.5),
         "clustered": lambda: wah.gen_clustered_device(n, 1337)}[kind]()
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    st = comp.result().clone()
    c = st.numel()
    ws_bytes = int(lib.wah_decompress_workspace_bytes(c, 0))
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
    info = torch.zeros(2, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.wah_decompress_scan_device(st.data_ptr(), c, info.data_ptr(), ws.data_ptr(), ws_bytes, s)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        lib.wah_decompress_scan_device(st.data_ptr(), c, info.data_ptr(), ws.data_ptr(), ws_bytes, s)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    assert int(info[1].item()) == (32 * n + 30) // 31
    print(f"  {kind}: C = {c} words, sums pass {ms*1e3:.1f} us -> {4*c/ms/1e6:.0f} GB/s")
    del d, comp, st, ws
