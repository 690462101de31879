"""AND of four 1 GiB bitmaps held compressed with their indexes: three pairwise wah_bitop_indexed_device calls against
one wah_bitop_many_indexed_device call."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
wah = importlib.import_module("gpu-wah_amd")
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


cap = wah.max_compressed_words(n)
nseg = (cap + 1023) // 1024
for name, gen in (("sparse p=0.05", lambda s: wah.gen_uniform_device(n, s, 0.05)), ("clustered", lambda s: wah.gen_clustered_device(n, s))):
    ops = [indexed(gen(s)) for s in (1, 2, 3, 4)]
    sc = torch.empty(int(wah.lib().wah_bitop_indexed_scratch_bytes(n)), dtype=torch.uint8, device="cuda")
    outs = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(2)]
    offs = [torch.zeros(nseg + 1, dtype=torch.int64, device="cuda") for _ in range(2)]

    def pairwise():
        wah.bitop_indexed_device("and", ops[0][0], ops[0][1], ops[1][0], ops[1][1], n, scratch=sc, out=outs[0], out_offsets=offs[0], check=False)
        wah.bitop_indexed_device("and", outs[0], offs[0], ops[2][0], ops[2][1], n, scratch=sc, out=outs[1], out_offsets=offs[1], check=False)
        return wah.bitop_indexed_device("and", outs[1], offs[1], ops[3][0], ops[3][1], n, scratch=sc, out=outs[0], out_offsets=offs[0], check=False)

    t_pair = timed(pairwise)
    _, cnt, _ = pairwise()
    torch.cuda.synchronize()
    ref = outs[0][: int(cnt.item())].clone()
    t_many = timed(lambda: wah.bitop_many_indexed_device("and", ops, n, scratch=sc, out=outs[1], out_offsets=offs[1], check=False))
    got, _ = wah.bitop_many_indexed_device("and", ops, n, scratch=sc, out=outs[1], out_offsets=offs[1])
    assert torch.equal(got, ref)
    print(f"{name}: operands {[o[0].numel() for o in ops]} words -> {ref.numel()}; three pairwise calls {t_pair:.3f} ms, one four-operand call {t_many:.3f} ms", flush=True)
    del ops, sc, outs, offs, ref, got
