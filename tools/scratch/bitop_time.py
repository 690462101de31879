"""AND of two 1 GiB bitmaps held compressed in HBM: time of wah_bitop_device (decode, decode, combine + compress)."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
lib = wah.lib()
n = 268435200
for name, gen in (("sparse p=0.01 & sparse", lambda s: wah.gen_uniform_device(n, s, 0.01)), ("clustered & clustered", lambda s: wah.gen_clustered_device(n, s))):
    a = wah.compress_device(gen(1)); b = wah.compress_device(gen(2))
    ca, cb = a.numel(), b.numel()
    cap = wah.max_compressed_words(n)
    sc_bytes = int(lib.wah_bitop_scratch_bytes(n, ca, cb))
    scratch = torch.empty(sc_bytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(cap, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: lib.wah_bitop_device(0, n, a.data_ptr(), ca, b.data_ptr(), cb, out.data_ptr(), cap, cnt.data_ptr(), scratch.data_ptr(), sc_bytes, s)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10): run()
    ev[1].record(); torch.cuda.synchronize()
    assert lib.wah_bitop_status(scratch.data_ptr(), n, ca, cb, s) == 0
    ms = ev[0].elapsed_time(ev[1]) / 10
    print(f"{name}: A {ca} + B {cb} words -> {int(cnt.item())} words, {ms:.3f} ms  ({4.0 * n / ms / 1e6:.0f} GB/s of bitmap per operand)", flush=True)
    del a, b, scratch, out
