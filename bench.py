#!/usr/bin/env python3
"""bench.py -- headline benchmark of the WAH hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one synthetic bitmap resident in HBM:
compress() followed by decompress() of the result (BASELINE.json configs[1]: 1 GiB
uniform p = 0.01, round trip on one MI355X).  `value` is uncompressed input bytes per
second through the whole round trip, aggregated over all ranks; with N > 1 every rank
works on its own independent column (no collective on the data path -- columns are
unrelated bitmaps), so scaling is "weak".

The JSON line also carries
  roofline     : the dominant kernel (compress) against the HBM roof.  achieved =
                 algorithmic bytes (4N + 4C) / average launch duration,
                 measured here with device events on the stream the kernel runs on.
  cpu_baseline : the CPU oracle (a port of the reference algorithm; the reference has
                 no CPU path and its CUDA cannot run here) timed on a bounded sample of
                 the same workload on this box's host cores.
Other workloads (--workload clustered|dense|sparse) are the remaining BASELINE configs.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)

WORKLOADS = {
    # name: (generator, parameter, description)
    "sparse": ("uniform", 0.01, "1 GiB uniform p=0.01 bitmap, compress+decompress round trip (BASELINE configs[1])"),
    "clustered": ("clustered", 4096, "1 GiB clustered runs (mean 4096 bits), compress+decompress (BASELINE configs[2])"),
    "dense": ("uniform", 0.5, "1 GiB uniform p=0.5 bitmap, all literals (BASELINE configs[3])"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sparse", choices=sorted(WORKLOADS))
    ap.add_argument("--words", type=int, default=268435200, help="bitmap size in 32-bit words (default: 270600 whole segments = 1 GiB - 256 words)")
    ap.add_argument("--cpu-sample-mib", type=int, default=256, help="size of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1337)
    return ap.parse_args()


def make_input(wah, kind, param, n, seed):
    if kind == "uniform":
        return wah.gen_uniform_device(n, seed, param)
    return wah.gen_clustered_device(n, seed, param)


def cpu_baseline(kind, param, sample_mib, seed):
    """Time the CPU oracle on a bounded sample of the workload (rank 0, N = 1 only)."""
    from tests import _oracle

    oracle = _oracle.load()
    n = (sample_mib << 20) // 4 // 992 * 992
    data = oracle.gen_uniform(n, seed, param) if kind == "uniform" else oracle.gen_clustered(n, seed, param)
    tc, td, _ = oracle.time_round_trip(data, threads=1, reps=2)
    nbytes = 4.0 * n
    res = {
        "value": round(nbytes / (tc + td) / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": f"{sample_mib} MiB of the same {kind} bitmap (seed {seed}), serial C oracle, compress+decompress, best of 2",
        "compress_GBps": round(nbytes / tc / 1e9, 4), "decompress_GBps": round(nbytes / td / 1e9, 4),
    }
    threads = os.cpu_count() or 1
    if threads > 1:
        tcm, _, _ = oracle.time_round_trip(data, threads=threads, reps=2)
        res["compress_all_cores_GBps"] = round(nbytes / tcm / 1e9, 4)
        res["all_cores"] = threads
    return res


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    wah = importlib.import_module("gpu-wah_amd")
    wah.lib()

    kind, param, desc = WORKLOADS[args.workload]
    n = args.words
    d_in = make_input(wah, kind, param, n, args.seed + rank)  # every rank: its own column
    comp = wah.DeviceCompressor(n, device=dev)
    comp.run(d_in)
    c_words = comp.result().numel()
    dec = wah.DeviceDecompressor(c_words, n + 1, device=dev)
    dec.run(comp.out)
    assert torch.equal(dec.result()[:n], d_in), "round trip mismatch"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(ev=None):
        if ev:
            ev[0].record()
        comp.run(d_in)
        if ev:
            ev[1].record()
        dec.run(comp.out, c_words)
        if ev:
            ev[2].record()

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for ev in events:
        step(ev)
    barrier()
    elapsed = time.perf_counter() - t0
    comp.status()
    dec.status()

    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    elapsed = float(t_all.item())

    comp_ms = sorted(e[0].elapsed_time(e[1]) for e in events)
    dec_ms = sorted(e[1].elapsed_time(e[2]) for e in events)
    comp_avg = sum(comp_ms) / len(comp_ms)
    dec_avg = sum(dec_ms) / len(dec_ms)

    # on-box copy ceiling: 16 B/lane copy of the same 1 GiB (read + write)
    scratch = torch.empty_like(d_in)
    for _ in range(2):
        wah.copy_device(d_in, scratch)
    ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ce[0].record()
    for _ in range(5):
        wah.copy_device(d_in, scratch)
    ce[1].record()
    torch.cuda.synchronize()
    copy_gbps = 5 * 8.0 * n / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9
    del scratch

    if rank == 0:
        in_bytes = 4.0 * n
        algo_c = 4.0 * n + 4.0 * c_words                       # compress: read N, write C
        algo_d = 4.0 * c_words + 4.0 * ((31 * ((32 * n + 30) // 31) + 31) // 32)  # decompress: read C, write N'
        achieved = algo_c / (comp_avg * 1e-3) / 1e9
        out = {
            "metric": "compress+decompress GB/s (input bits), 1 GiB bitmap",
            "value": round(world * args.steps * in_bytes / elapsed / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": desc, "words": n, "seed": args.seed, "columns_per_gpu": 1,
                       "parallelism": f"column-shard x{world}, no collective"},
            "compression_ratio_C_over_N": round(c_words / n, 6),
            "compress_GBps": round(in_bytes / (comp_avg * 1e-3) / 1e9, 2),
            "decompress_GBps": round(in_bytes / (dec_avg * 1e-3) / 1e9, 2),
            "compress_ms": {"avg": round(comp_avg, 4), "min": round(comp_ms[0], 4), "median": round(comp_ms[len(comp_ms) // 2], 4)},
            "decompress_ms": {"avg": round(dec_avg, 4), "min": round(dec_ms[0], 4), "median": round(dec_ms[len(dec_ms) // 2], 4)},
            "copy_ceiling_GBps": round(copy_gbps, 1),
            "roofline": {"kernel": "compress_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": algo_c, "frac_of_copy_ceiling": round(achieved / copy_gbps, 4)},
            "roofline_decompress": {"kernel": "decode_scan_kernel+decode_expand_kernel", "bound": "hbm",
                                    "achieved": round(algo_d / (dec_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                    "unit": "GB/s", "frac": round(algo_d / (dec_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                    "traffic": None, "algorithmic_bytes_per_launch": algo_d},
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file)).get(args.workload)
                if tr:
                    out["roofline"]["traffic"] = tr.get("compress_bytes_per_launch")
                    out["roofline"]["traffic_source"] = tr.get("source")
                    out["roofline_decompress"]["traffic"] = tr.get("decompress_bytes_per_launch")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, param, args.cpu_sample_mib, args.seed)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
