"""AND of two 1 GiB bitmaps held compressed in HBM: wah_bitop_device (decode, decode, combine + compress) against
wah_bitop_indexed_device (one combining pass through the segment indexes + compress)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wah = importlib.import_module("gpu-wah_amd")
lib = wah.lib()
n = 268435200


def timed(run, reps=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def indexed(d):
    c = wah.DeviceCompressor(n, indexed=True)
    c.run(d)
    return c.result().clone(), c.seg_offsets.clone()


for name, gen in (("sparse p=0.01 & sparse", lambda s: wah.gen_uniform_device(n, s, 0.01)),
                  ("clustered & clustered", lambda s: wah.gen_clustered_device(n, s)),
                  ("dense p=0.5 & sparse", lambda s: wah.gen_uniform_device(n, s, 0.5 if s == 1 else 0.01))):
    (a, oa), (b, ob) = indexed(gen(1)), indexed(gen(2))
    ca, cb = a.numel(), b.numel()
    cap = wah.max_compressed_words(n)
    sc_bytes = int(lib.wah_bitop_scratch_bytes(n, ca, cb))
    scratch = torch.empty(sc_bytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ms = timed(lambda: lib.wah_bitop_device(0, n, a.data_ptr(), ca, b.data_ptr(), cb, out.data_ptr(), cap, cnt.data_ptr(), scratch.data_ptr(), sc_bytes, s))
    assert lib.wah_bitop_status(scratch.data_ptr(), n, ca, cb, s) == 0
    ref = out[: int(cnt.item())].clone()
    del scratch
    isc = torch.empty(int(lib.wah_bitop_indexed_scratch_bytes(n)), dtype=torch.uint8, device="cuda")
    ooffs = torch.zeros((cap + 1023) // 1024 + 1, dtype=torch.int64, device="cuda")
    ims = timed(lambda: wah.bitop_indexed_device("and", a, oa, b, ob, n, scratch=isc, out=out, out_offsets=ooffs, check=False))
    assert lib.wah_bitop_indexed_status(isc.data_ptr(), n, s) == 0
    got, _ = wah.bitop_indexed_device("and", a, oa, b, ob, n, scratch=isc, out=out, out_offsets=ooffs)
    assert torch.equal(got, ref)
    print(f"{name}: A {ca} + B {cb} words -> {ref.numel()} words; general {ms:.3f} ms, indexed {ims:.3f} ms "
          f"({4.0 * n / ims / 1e6:.0f} GB/s of bitmap per operand)", flush=True)
    del a, b, oa, ob, out, isc, ref, got
