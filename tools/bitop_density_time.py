"""The two routes of the indexed bit operations (include/wah.h: WAH_BITOP_ROUTE_RUNS / _GROUPS) by what the operands hold: AND of
two and OR of four and of eight 1 GiB bitmaps held compressed in HBM -- uniform, one bit in 2^i, and clustered with mean runs of r bits.
Needs the experiment build (make -C gpu-wah_amd exp: WAH_BITOP_ROUTE forces a route); both results are compared.
usage: python tools/bitop_density_time.py [uI | cR ...]   (default u5 u7 u9 u11 u13 c1024 c4096 c16384)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WAH_LIB_PATH", os.path.join(ROOT, "gpu-wah_amd", "libwah_hip_exp.so"))
import torch  # noqa: E402
wah = importlib.import_module("gpu-wah_amd")
lib = wah.lib()
n = 268435200


def timed(run, reps=5):
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
n_seg = (cap + 1023) // 1024
isc = torch.empty(int(lib.wah_bitop_indexed_scratch_bytes(n)), dtype=torch.uint8, device="cuda")
out = torch.empty(cap, dtype=torch.int32, device="cuda")
ooffs = torch.zeros(n_seg + 1, dtype=torch.int64, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
for kind in sys.argv[1:] or ["u5", "u7", "u9", "u11", "u13", "c1024", "c4096", "c16384"]:
    x = int(kind[1:])
    gen = (lambda s: wah.gen_uniform_device(n, s, 2.0 ** -x)) if kind[0] == "u" else (lambda s: wah.gen_clustered_device(n, s, x))
    ops = [indexed(gen(s)) for s in range(1, 9)]
    words = [int(o[0].numel()) for o in ops]
    row = []
    for label, run in (("AND of 2", lambda: wah.bitop_indexed_device("and", *ops[0], *ops[1], n, scratch=isc, out=out, out_offsets=ooffs, check=False)),
                       ("OR of 4", lambda: wah.bitop_many_indexed_device("or", ops[:4], n, scratch=isc, out=out, out_offsets=ooffs, check=False)),
                       ("OR of 8", lambda: wah.bitop_many_indexed_device("or", ops, n, scratch=isc, out=out, out_offsets=ooffs, check=False))):
        res = {}
        for route in ("groups", "runs"):
            os.environ["WAH_BITOP_ROUTE"] = route
            ms = timed(run)
            assert lib.wah_bitop_indexed_status(isc.data_ptr(), n, sp) == 0
            o, c, _ = run()
            torch.cuda.synchronize()
            res[route] = (ms, o[: int(c.item())].clone(), ooffs.clone())
        same = torch.equal(res["groups"][1], res["runs"][1]) and torch.equal(res["groups"][2], res["runs"][2])
        row.append(f"{label}: groups {res['groups'][0]:.3f} ms, runs {res['runs'][0]:.3f} ms -> {res['runs'][1].numel()} words{'' if same else '  RESULTS DIFFER'}")
        del res
    per_seg = sum(words[:2]) / n_seg
    print(f"{kind}: {words[0]} words per operand ({per_seg:.0f} per segment for two, {sum(words[:4]) / n_seg:.0f} for four, {sum(words) / n_seg:.0f} for eight)   " + "   ".join(row), flush=True)
    del ops
