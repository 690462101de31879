"""The kernels of SURVEY.md 8(f) -- unsegmented encoder, stream checker, index builder, fill merger, the bit operations on
compressed bitmaps -- on the 1 GiB sparse and clustered bitmaps: every call five times back to back on one stream.  Run under
`rocprofv3 --kernel-trace` (tools/make_profiles.sh) the kernel trace gives each kernel's launch time; this script prints, per
workload, the ALGORITHMIC bytes of every kernel as one JSON line (`NEXT_ROWS {...}`) so that the summary can turn the times into
fractions of the 8 TB/s roofline, and the wall time per call between two events.
usage: python tools/next_rows_time.py [sparse|clustered ...]"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
lib = wah.lib()
n = 268435200
REPS = 5


def timed(run):
    run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(REPS):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / REPS


def indexed(d):
    c = wah.DeviceCompressor(n, indexed=True)
    c.run(d)
    return c.result().clone(), c.seg_offsets.clone()


for kind in sys.argv[1:] or ["sparse", "clustered"]:
    gen = (lambda s: wah.gen_uniform_device(n, s, 0.01)) if kind == "sparse" else (lambda s: wah.gen_clustered_device(n, s))
    d = gen(1337)
    groups = wah.max_compressed_words(n)
    n_seg = (groups + 1023) // 1024
    rows, calls = {}, {}
    sp = torch.cuda.current_stream().cuda_stream

    # ---- the unsegmented encoder mode (f.3): read N, write C_u
    un = wah.DeviceCompressor(n, unsegmented=True)
    calls["wah_compress_device_ex(WAH_UNSEGMENTED)"] = timed(lambda: un.run(d))
    c_un = int(un.result().numel())
    rows["compress_unseg_pair_kernel"] = 4.0 * n + 4.0 * c_un

    # ---- a segmented stream and its index
    a, oa = indexed(d)
    ca = int(a.numel())

    # ---- stream checker (f.3): the scan of the stream (4C) + one position-aware pass (4C)
    ws = torch.zeros(int(lib.wah_decompress_workspace_bytes(ca, 0)), dtype=torch.uint8, device="cuda")
    report = torch.zeros(8, dtype=torch.int64, device="cuda")
    calls["wah_validate_device"] = timed(lambda: lib.wah_validate_device(a.data_ptr(), ca, report.data_ptr(), ws.data_ptr(), ws.numel(), sp))
    assert lib.wah_decompress_status(ws.data_ptr(), sp) == 0 and int(report[6].item()) == 1
    rows["validate_kernel"] = 4.0 * ca
    rows["decode_sums_kernel"] = 4.0 * ca

    # ---- the index of a stream that came without one (f.1): scan (4C) + one pass that places every word (4C read, 8 per segment written)
    offs = torch.zeros(ca + 1, dtype=torch.int64, device="cuda")
    info = torch.zeros(2, dtype=torch.int64, device="cuda")
    calls["wah_build_index_device"] = timed(lambda: lib.wah_build_index_device(a.data_ptr(), ca, offs.data_ptr(), offs.numel(), info.data_ptr(),
                                                                               ws.data_ptr(), ws.numel(), sp))
    assert lib.wah_decompress_status(ws.data_ptr(), sp) == 0 and torch.equal(offs[: n_seg + 1], oa[: n_seg + 1])
    rows["index_kernel"] = 4.0 * ca + 8.0 * (n_seg + 1)
    del offs

    # ---- fill merger (f.3): scan (4C), kept words per tile (4C read), scan of the tiles, scatter (4C read + 8C positions + 4C_u written), fix-up
    mws = torch.zeros(int(lib.wah_merge_fills_workspace_bytes(ca)), dtype=torch.uint8, device="cuda")
    mout = torch.empty(ca, dtype=torch.int32, device="cuda")
    mcnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    calls["wah_merge_fills_device"] = timed(lambda: lib.wah_merge_fills_device(a.data_ptr(), ca, mout.data_ptr(), ca, mcnt.data_ptr(), mws.data_ptr(),
                                                                               mws.numel(), sp))
    assert lib.wah_decompress_status(mws.data_ptr(), sp) == 0 and int(mcnt.item()) == c_un
    rows["merge_count_kernel"] = 4.0 * ca
    rows["merge_scatter_kernel"] = 4.0 * ca + 4.0 * c_un
    del mws, mout

    # ---- bit operations (f.4): AND of two indexed streams, fused (read C_A + C_B, write C_out)
    b, ob = indexed(gen(4242))
    cb = int(b.numel())
    isc = torch.empty(int(lib.wah_bitop_indexed_scratch_bytes(n)), dtype=torch.uint8, device="cuda")
    out = torch.empty(groups, dtype=torch.int32, device="cuda")
    ooffs = torch.zeros(n_seg + 1, dtype=torch.int64, device="cuda")
    calls["wah_bitop_indexed_device(AND, 2 operands)"] = timed(lambda: wah.bitop_indexed_device("and", a, oa, b, ob, n, scratch=isc, out=out, out_offsets=ooffs, check=False))
    assert lib.wah_bitop_indexed_status(isc.data_ptr(), n, sp) == 0
    got, _ = wah.bitop_indexed_device("and", a, oa, b, ob, n, scratch=isc, out=out, out_offsets=ooffs)
    if lib.wah_last_bitop_route() == 1:  # operands of few words per segment: their runs merged (wah_bitop_runs.hip), then moved together
        rows["bitop_runs_kernel<2"] = 4.0 * ca + 4.0 * cb + 4.0 * int(got.numel())
        rows["bitop_runs_place_kernel<2"] = 8.0 * int(got.numel())
    else:
        rows["bitop_tile_kernel"] = 4.0 * ca + 4.0 * cb + 4.0 * int(got.numel())

    # ---- ... of four: one combining pass (read the four streams, write one decoded bitmap) + the compress kernel over that bitmap
    ops = [(a, oa), (b, ob), indexed(gen(99)), indexed(gen(7))]
    calls["wah_bitop_many_indexed_device(AND, 4 operands)"] = timed(lambda: wah.bitop_many_indexed_device("and", ops, n, scratch=isc, out=out, out_offsets=ooffs, check=False))
    assert lib.wah_bitop_indexed_status(isc.data_ptr(), n, sp) == 0
    got4, _ = wah.bitop_many_indexed_device("and", ops, n, scratch=isc, out=out, out_offsets=ooffs)
    if lib.wah_last_bitop_route() == 1:
        rows["bitop_runs_kernel<4"] = 4.0 * sum(int(s.numel()) for s, _ in ops) + 4.0 * int(got4.numel())
        rows["bitop_runs_place_kernel<4"] = 8.0 * int(got4.numel())
    else:
        rows["bitop_many_segments_kernel"] = 4.0 * sum(int(s.numel()) for s, _ in ops) + 4.0 * wah.decoded_words(groups)
        rows["compress_pair_kernel"] = 4.0 * n + 4.0 * int(got4.numel())  # (its launch inside the four-operand call)

    print("NEXT_ROWS " + json.dumps({"workload": kind, "n_words": n, "c_words": ca, "c_unsegmented": c_un, "algorithmic_bytes": rows,
                                     "call_ms": {k: round(v, 4) for k, v in calls.items()}}), flush=True)
    del a, b, oa, ob, ops, isc, out, ooffs, d, un, ws
    torch.cuda.empty_cache()
