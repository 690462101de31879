"""Does a second pass over the same data come out of the memory-side cache?  Times the sums pass (reads C once) and the
plain copy kernel over buffers of growing size, repeatedly: bandwidth against footprint."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
lib = wah.lib()
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024):
    n = mb * 1024 * 1024 // 4
    d = wah.gen_uniform_device(n, 7, 0.5)  # any words: the sums pass reads them as a "stream"
    c = int(n)
    ws_bytes = int(lib.wah_decompress_workspace_bytes(c, 0))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    info = torch.zeros(2, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.wah_decompress_scan_device(d.data_ptr(), c, info.data_ptr(), ws.data_ptr(), ws_bytes, s)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 20
    ev[0].record()
    for _ in range(reps):
        lib.wah_decompress_scan_device(d.data_ptr(), c, info.data_ptr(), ws.data_ptr(), ws_bytes, s)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    print(f"{mb:5d} MB: sums pass {ms * 1e3:8.1f} us  -> {mb * 1.048576 / ms:8.1f} GB/s", flush=True)
    del d, ws
