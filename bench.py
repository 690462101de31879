#!/usr/bin/env python3
"""bench.py -- headline benchmark of the WAH hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sparse|clustered|dense|columns]

N = 1 runs in this process.  N > 1: when the process was not started by a launcher (no WORLD_SIZE in the environment)
it starts N fresh ranks itself -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
... bench.py <same arguments>` -- BEFORE it touches a GPU, relays rank 0's JSON line and exits with the ranks' status;
started by a launcher it checks that WORLD_SIZE equals --gpus and refuses to run otherwise.  It never prints a line
for fewer GPUs than were asked for.

One "step" = one pass of the hot path over synthetic bitmaps already resident in HBM:
  sparse / clustered / dense : compress() then decompress() of one 1 GiB bitmap per GPU
                               (BASELINE.json configs[1] / [2] / [3]; default: sparse = configs[1]); every rank has its
                               own bitmap, so per-GPU work is fixed: "scaling": "weak"
  columns                    : compress() of 1024 independent 128 MiB columns (configs[4]), column c on rank c mod N,
                               resident in HBM, compressed in batches of up to 128 columns per launch; the total is
                               fixed, so "scaling": "strong"
`value` is uncompressed input bytes per second through the step, whole job; the time is the maximum over ranks of a
barrier-bracketed wall clock.  There is no collective on the data path and RCCL is not initialised at all: the two
scalars of the report (bytes, slowest rank's time) and the barriers go over gloo on 127.0.0.1.

The JSON line also carries
  roofline     : the dominant kernel (compress_pair_kernel) against the HBM roof.  achieved = algorithmic bytes per
                 launch (4N + 4C) / its average launch duration inside the round trip, measured
                 here with device events on the stream it runs on; launch_ms_isolated = the same kernel in a
                 compress-only loop (in the round trip it starts while the expand kernel's 1 GiB of writes is still
                 draining from the memory-side cache).  `traffic` = HBM bytes per launch by the PMC counters (rocprofv3
                 FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes + WRITE_SIZE, separate passes), counted IN THIS RUN
                 by two child runs of the same command (3 steps) under rocprofv3 before the timed process touches the GPU
                 ("traffic_measured_in_run": true); without rocprofv3, with --no-traffic, or when a pass fails: the number
                 recorded in profiles/traffic.json, replayed ("traffic_measured_in_run": false).
  columns      : side block (never `value`): BASELINE configs[4] measured by the same command -- this rank's share of
                 the 1024 independent 128 MiB columns, 5 steps -- with its own roofline block; --no-columns leaves it out.
  cpu_baseline : the CPU oracle (a port of the reference algorithm -- the reference has no CPU path and its CUDA
                 cannot run here) timed on a bounded sample of the same workload on this box's host cores.

WAH_BENCH_REHEARSE=1   : rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share the cards).
WAH_BENCH_REHEARSE=cpu : rehearsal without any GPU: launcher, rendezvous, sharding, reductions and the JSON line with
                         empty steps.  Either way the line says "rehearsal" and its numbers mean nothing.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
COLUMN_WORDS = 33554400  # 128 MiB - 128 B: 33825 whole 992-word segments
COLUMNS_TOTAL = 1024     # BASELINE.json configs[4]
COLUMNS_PER_LAUNCH = 128  # 16 GiB of columns per launch: input + worst-case output of a batch stay far below HBM

WORKLOADS = {
    # name: (generator, parameter, description)
    "sparse": ("uniform", 0.01, "1 GiB uniform p=0.01 bitmap, compress+decompress round trip (BASELINE configs[1])"),
    "clustered": ("clustered", 4096, "1 GiB clustered runs (mean 4096 bits), compress+decompress (BASELINE configs[2])"),
    "dense": ("uniform", 0.5, "1 GiB uniform p=0.5 bitmap, all literals, compress+decompress (BASELINE configs[3])"),
    "columns": (None, None, "1024 independent 128 MiB bitmap columns (sparse/clustered/dense in turn), compress, column c "
                            "on rank c mod N (BASELINE configs[4])"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sparse", choices=sorted(WORKLOADS))
    ap.add_argument("--words", type=int, default=None,
                    help="bitmap size in 32-bit words (default 268435200 = 270600 whole segments = 1 GiB - 1 KiB; "
                         f"columns: {COLUMN_WORDS} = 128 MiB)")
    ap.add_argument("--columns", type=int, default=None, help=f"columns workload: total number of columns (default {COLUMNS_TOTAL})")
    ap.add_argument("--columns-per-launch", type=int, default=COLUMNS_PER_LAUNCH)
    ap.add_argument("--per-column-launches", action="store_true", help="columns workload: one launch per column")
    ap.add_argument("--two-in-flight", action="store_true",
                    help="round-trip workloads: also measure the throughput with two round trips in flight on two streams "
                         "(a side field; off by default so that a kernel trace of the default command holds one launch at a time)")
    ap.add_argument("--streams", type=int, default=2,
                    help="columns workload: HIP streams the launches are spread over (one compressor = output buffer + workspace each)")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024, help="size of the CPU-baseline sample (default: the whole 1 GiB bitmap, about 10 s of host work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not count HBM bytes with rocprofv3 child runs (roofline.traffic is then replayed from profiles/traffic.json)")
    ap.add_argument("--no-columns", action="store_true", help="round-trip workloads: leave out the `columns` side block (configs[4])")
    ap.add_argument("--columns-steps", type=int, default=5, help="steps of the `columns` side block")
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--master-port", type=int, default=None, help="rendezvous port when this process starts the ranks itself")
    ap.add_argument("--dry-launch", action="store_true", help="print the launcher command this call would run and exit")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks
# ----------------------------------------------------------------------------------------------------------------
def launcher_command(args, argv):
    """The command that starts args.gpus ranks of this script (one per GPU) on this node."""
    port = args.master_port or (29500 + os.getpid() % 2000)
    passed = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + passed


def launch_ranks(args, argv):
    """Parent of an N > 1 run: no GPU call is made in this process.  Returns the exit status."""
    cmd = launcher_command(args, argv)
    if args.dry_launch:
        print(json.dumps({"dry_launch": cmd, "n_gpus": args.gpus}), flush=True)
        return 0
    rehearse = os.environ.get("WAH_BENCH_REHEARSE")
    if not rehearse:
        import torch  # device_count() does not initialise the GPU

        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to report a smaller run "
                  f"(WAH_BENCH_REHEARSE=1 rehearses the multi-rank path on shared cards)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stderr.write(proc.stdout)
        print(f"bench.py: the {args.gpus}-rank run failed (exit {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    if json.loads(line).get("n_gpus") != args.gpus:
        print("bench.py: the ranks reported a different GPU count than requested", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------------------
# measurement
# ----------------------------------------------------------------------------------------------------------------
def timed_steps(step, steps, warmup, dist=None, device=None):
    """The measurement contract: W untimed steps, then exactly K steps between two barriers (+ device syncs), and
    the MAX over ranks of the wall time.  `dist` is torch.distributed (or None), `device` a CUDA device (or None on
    the CPU rehearsal)."""
    import torch

    def barrier():
        if device is not None:
            torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        if device is not None:
            torch.cuda.synchronize(device)

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def cpu_baseline(kind, param, sample_mib, seed):
    """Time the CPU oracle on a bounded sample of the workload (rank 0, N = 1 only)."""
    from tests import _oracle

    oracle = _oracle.load()
    n = (sample_mib << 20) // 4 // 992 * 992
    data = oracle.gen_uniform(n, seed, param) if kind == "uniform" else oracle.gen_clustered(n, seed, param)
    tc, td, _ = oracle.time_round_trip(data, threads=1, reps=2)
    nbytes = 4.0 * n
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((line.split(":", 1)[1].strip() for line in f if line.startswith("model name")), None)
    except OSError:
        pass
    res = {
        "value": round(nbytes / (tc + td) / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
        "sample": f"{sample_mib} MiB of the same {kind} bitmap (seed {seed}), serial C oracle, compress+decompress, best of 2",
        "compress_GBps": round(nbytes / tc / 1e9, 4), "decompress_GBps": round(nbytes / td / 1e9, 4),
    }
    threads = os.cpu_count() or 1
    if threads > 1:
        tcm, _, _ = oracle.time_round_trip(data, threads=threads, reps=2)
        res["compress_all_cores_GBps"] = round(nbytes / tcm / 1e9, 4)
        res["all_cores"] = threads
    return res


def load_traffic(workload):
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(workload)
    except Exception:
        return None


def measure_traffic(args):
    """HBM bytes per launch of the round trip's kernels, counted IN THIS RUN: two child runs of this command (3 steps) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes;
    FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 bytes), started before this process touches the GPU.
    None when rocprofv3 is not there or a pass fails (the line then replays profiles/traffic.json and says so)."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import tempfile

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None
    # (already under a profiler -- its preloaded library has initialised the GPU in THIS process: no child may be started from it)
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY")) or \
            any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in os.environ):
        return None
    raw = collections.defaultdict(dict)
    tmp = tempfile.mkdtemp(prefix="wah_traffic_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-columns",
                   "--no-traffic", "--workload", args.workload, "--seed", str(args.seed)]
            if args.words:
                cmd += ["--words", str(args.words)]
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            if p.returncode != 0:
                return None
            agg = collections.defaultdict(list)
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                    if m and r["Counter_Name"] == counter:
                        agg[m.group(1)].append(float(r["Counter_Value"]))
            if not agg:
                return None
            for k, v in agg.items():
                raw[k][counter] = sum(v) / len(v)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def hbm(k):
        t = raw.get(k, {})
        return (2.0 * t.get("FETCH_SIZE", 0.0) + t.get("WRITE_SIZE", 0.0)) * 1024.0  # (the counters are in KiB)

    return {"compress_bytes_per_launch": hbm("compress_pair_kernel"),
            "decompress_bytes_per_launch": hbm("decode_tile_kernel") + hbm("decode_expand_list_kernel") + hbm("decode_sums_kernel") + hbm("decode_expand_kernel"),
            "decompress_indexed_bytes_per_launch": hbm("decode_segments_kernel"),
            "source": "this run: two child runs of the same command (3 steps) under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE "
                      "(separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md; mean per launch",
            "measured": True}


def column_plan(n_columns, rank, world, per_launch):
    """This rank's columns (c with c % world == rank) and how they are cut into launches."""
    mine = list(range(rank, n_columns, world))
    batches = [mine[i:i + per_launch] for i in range(0, len(mine), max(per_launch, 1))]
    return mine, batches


def columns_measurement(args, dev, rank, world, dist, cpu_only, steps, warmup, desc):
    """BASELINE configs[4]: compress() of this rank's share of the independent 128 MiB columns (column c on rank c mod N),
    resident in HBM, `columns_per_launch` columns per launch, the launches round robin over a few streams.  Returns the
    JSON object of the measurement on rank 0 (None elsewhere).  Every launch is checked once, outside the timed region,
    one launch at a time (status + a C that fits its buffer); inside the timed region the launches of a stream write
    their streams into that stream's output buffer one after the other -- nothing consumes them, the figure is the
    compressor's throughput."""
    import torch

    n = args.words if (args.words and args.workload == "columns") else COLUMN_WORDS
    n_columns = args.columns or COLUMNS_TOTAL
    mine, batches = column_plan(n_columns, rank, world, 1 if args.per_column_launches else args.columns_per_launch)
    c_words_rank = 0.0
    one_stream_ms = None
    n_streams = 1
    if cpu_only:
        def step():
            pass
    else:
        wah = importlib.import_module("gpu-wah_amd")
        wah.lib()
        specs = [wah.columns.column_spec(c, n, args.seed) for c in mine]
        batched = n % wah.columns.SEGMENT_WORDS == 0 and not args.per_column_launches
        # all of the rank's columns stay resident: one [columns, n] matrix, generated once
        matrix = wah.columns.make_column_matrix(wah, specs, dev) if specs else None
        widest = max((len(b) for b in batches), default=1)
        # the launches go round robin over a few streams, each with its own compressor (output buffer + workspace):
        # the tail of one launch overlaps the start of the next ("just per-GPU streams", north_star)
        n_streams = max(1, min(args.streams, len(batches)))
        comps = [wah.DeviceCompressor(widest * n if batched else n, device=dev, indexed=batched) for _ in range(n_streams)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [None]
        sizes = torch.zeros(max(len(batches), 1), dtype=torch.int64, device=dev)  # C of every launch, written by the kernels

        def launch(i, b, row, comp, st):
            rows = matrix[row:row + len(b)]
            if batched:
                # ONE launch per batch: whole 992-word segments, so the result is the columns' streams back to back;
                # the segment index gives the column boundaries
                comp.run(rows.view(-1), n_words=rows.numel(), stream=st, count=sizes[i:i + 1])
            else:
                comp.run(rows[0], stream=st, count=sizes[i:i + 1])

        def step():
            row = 0
            main = torch.cuda.current_stream(dev)
            for st in streams:
                if st is not None:
                    st.wait_stream(main)
            for i, b in enumerate(batches):
                launch(i, b, row, comps[i % n_streams], streams[i % n_streams])
                row += len(b)
            for st in streams:
                if st is not None:
                    main.wait_stream(st)

        # every launch once, on its own, with its status read back: an error of ANY launch is seen (inside the timed
        # loop a later launch on the same compressor starts by clearing the error word), and its time is the kernel's
        row, ev_ms = 0, []
        for i, b in enumerate(batches):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            launch(i, b, row, comps[i % n_streams], None)
            ev[1].record()
            torch.cuda.synchronize(dev)
            comps[i % n_streams].status()
            ev_ms.append(ev[0].elapsed_time(ev[1]))
            row += len(b)
        one_stream_ms = sum(ev_ms) / max(len(ev_ms), 1)
        c_words_rank = float(sizes.sum().item())
    elapsed = timed_steps(step, steps, warmup, dist, dev)
    if not cpu_only:
        for comp in comps:
            comp.status()
    stats = torch.tensor([4.0 * n * len(mine), c_words_rank], dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(stats)
    if rank != 0:
        return None
    in_bytes_job, c_words_job = float(stats[0].item()), float(stats[1].item())
    algo = in_bytes_job + 4.0 * c_words_job  # compress: read N, write C, all columns
    achieved = steps * algo / elapsed / 1e9
    return {
        "metric": "compress GB/s (input bits), 1024 independent 128 MiB bitmap columns",
        "value": round(steps * in_bytes_job / elapsed / 1e9, 3), "unit": "GB/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": desc, "words_per_column": n, "columns": n_columns,
                   "columns_per_gpu": len(mine), "seed": args.seed, "launches_per_step": len(batches),
                   "columns_per_launch": max((len(b) for b in batches), default=0),
                   "streams": n_streams,
                   "outputs": "every launch writes its stream into its HIP stream's output buffer (one per stream, reused by "
                              "the next launch on it): compressor throughput, nothing consumes the compressed columns",
                   "parallelism": f"column-shard x{world}, no collective"},
        "compression_ratio_C_over_N": round(c_words_job * 4.0 / in_bytes_job, 6) if in_bytes_job else None,
        "roofline": {"kernel": "compress_pair_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS * world,
                     "unit": "GB/s", "frac": round(achieved / (HBM_PEAK_GBPS * world), 4), "traffic": None,
                     "algorithmic_bytes_per_step": algo,
                     "note": "(4N + 4C) of all columns / wall time of the step, whole job against N x 8 TB/s; the launches of a "
                             "step overlap on the streams, so this is not one launch's duration",
                     "launch_ms_alone": round(one_stream_ms, 4) if one_stream_ms is not None else None,
                     "frac_launch_alone": (round(algo / max(len(batches), 1) / world / (one_stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                           if one_stream_ms else None)},
    }


def init_ranks(rank, world):
    """Control plane of an N > 1 run: gloo on 127.0.0.1 (barriers + two scalars).  RCCL is not used."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def run_rank(args):
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rehearse = os.environ.get("WAH_BENCH_REHEARSE")
    cpu_only = rehearse == "cpu"
    if not cpu_only and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = None
    if not cpu_only:
        if rehearse:
            local_rank %= torch.cuda.device_count()
        elif local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node")
        torch.cuda.set_device(local_rank)
        dev = torch.device(f"cuda:{local_rank}")
    dist = init_ranks(rank, world) if world > 1 else None
    kind, param, desc = WORKLOADS[args.workload]
    extra = {"rehearsal": f"WAH_BENCH_REHEARSE={rehearse}: numbers mean nothing"} if rehearse else {}

    def finish():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    # ---- independent columns: compress only, fixed total -----------------------------------------------------------
    if args.workload == "columns":
        out = columns_measurement(args, dev, rank, world, dist, cpu_only, args.steps, args.warmup, desc)
        if rank == 0:
            out.update(extra)
            print(json.dumps(out), flush=True)
        finish()
        return

    # ---- single bitmap per GPU: compress + decompress round trip ---------------------------------------------
    n = args.words or 268435200
    if cpu_only:
        elapsed = timed_steps(lambda: None, args.steps, args.warmup, dist, None)
        if rank == 0:
            out = {"metric": "compress+decompress GB/s (input bits), 1 GiB bitmap", "value": None, "unit": "GB/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                   "config": {"workload": desc, "words": n, "seed": args.seed, "bitmaps_per_gpu": 1,
                              "parallelism": f"column-shard x{world}, no collective"}}
            out.update(extra)
            print(json.dumps(out), flush=True)
        finish()
        return

    wah = importlib.import_module("gpu-wah_amd")
    wah.lib()
    d_in = (wah.gen_uniform_device(n, args.seed + rank, param, device=dev) if kind == "uniform"
            else wah.gen_clustered_device(n, args.seed + rank, param, device=dev))  # every rank: its own bitmap
    comp = wah.DeviceCompressor(n, device=dev)
    comp.run(d_in)
    c_words = comp.result().numel()
    dec = wah.DeviceDecompressor(c_words, n + 1, device=dev)
    dec.run(comp.out)
    assert torch.equal(dec.result()[:n], d_in), "round trip mismatch"

    events = []

    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        comp.run(d_in)
        ev[1].record()
        dec.run(comp.out, c_words)
        ev[2].record()
        events.append(ev)

    elapsed = timed_steps(step, args.steps, args.warmup, dist, dev)
    comp.status()
    dec.status()
    timed = events[args.warmup:]
    comp_ms = sorted(e[0].elapsed_time(e[1]) for e in timed)
    dec_ms = sorted(e[1].elapsed_time(e[2]) for e in timed)
    comp_avg = sum(comp_ms) / len(comp_ms)
    dec_avg = sum(dec_ms) / len(dec_ms)

    # the compress kernel on its own: a compress-only loop (no expand kernel's write drain in front of it)
    for _ in range(2):
        comp.run(d_in)
    ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ce[0].record()
    for _ in range(10):
        comp.run(d_in)
    ce[1].record()
    torch.cuda.synchronize()
    comp_isolated = ce[0].elapsed_time(ce[1]) / 10
    comp.status()

    # throughput with TWO round trips in flight (a side measurement, never `value`): a second compressor / decompressor
    # pair on a second stream works on the same bitmap, so the tail of one launch overlaps the start of the next
    # (independent requests of a serving system; `value` stays the one-at-a-time rate of the contract)
    two_in_flight = None
    if world == 1 and args.two_in_flight:
        comp2 = wah.DeviceCompressor(n, device=dev)
        dec2 = wah.DeviceDecompressor(c_words, n + 1, device=dev)
        side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)

        def pair_step():
            side.wait_stream(main)
            comp.run(d_in)
            dec.run(comp.out, c_words)
            comp2.run(d_in, stream=side)
            dec2.run(comp2.out, c_words, stream=side)
            main.wait_stream(side)

        pair_step()
        torch.cuda.synchronize(dev)
        assert torch.equal(dec2.result()[:n], d_in), "round trip mismatch (second stream)"
        pairs = max(args.steps // 2, 2)
        t0 = time.perf_counter()
        for _ in range(pairs):
            pair_step()
        torch.cuda.synchronize(dev)
        two_in_flight = 2 * pairs * 4.0 * n / (time.perf_counter() - t0) / 1e9
        comp2.status()
        dec2.status()
        del comp2, dec2

    # on-box copy ceiling: 16 B/lane copy of the same bitmap (read + write)
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

    # side measurement, not part of `value`: the same bitmap decoded through the compressor's segment index
    # (wah_decompress_segments_device: one pass, no sums kernel) -- what an engine that keeps the index gets
    icomp = wah.DeviceCompressor(n, device=dev, indexed=True)
    icomp.run(d_in)
    istream = icomp.result()
    seg_ws = torch.empty(int(wah.lib().wah_decompress_segments_workspace_bytes()), dtype=torch.uint8, device=dev)
    for _ in range(2):
        back = wah.decompress_segments_device(istream, icomp.seg_offsets, n, out=dec.out, workspace=seg_ws, check=False)
    ie = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ie[0].record()
    for _ in range(5):
        wah.decompress_segments_device(istream, icomp.seg_offsets, n, out=dec.out, workspace=seg_ws, check=False)
    ie[1].record()
    torch.cuda.synchronize()
    idx_ms = ie[0].elapsed_time(ie[1]) / 5
    assert torch.equal(back[:n], d_in), "index decode mismatch"
    del icomp, istream, back

    if rank == 0:
        in_bytes = 4.0 * n
        groups = (32 * n + 30) // 31
        algo_c = 4.0 * n + 4.0 * c_words                          # compress: read N, write C
        algo_d = 4.0 * c_words + 4.0 * ((31 * groups + 31) // 32)  # decompress: read C, write N'
        achieved = algo_c / (comp_avg * 1e-3) / 1e9
        achieved_d = algo_d / (dec_avg * 1e-3) / 1e9
        # the general decoder's route, as the LIBRARY reports it for the timed calls (wah_last_decode_route)
        decode_kernels = {"one pass": "decode_tile_kernel + decode_expand_list_kernel (the tiles it puts on its list: a highly "
                                      "compressed stream's all, an incompressible one's none)",
                          "two launches": "decode_sums_kernel + decode_expand_kernel",
                          "no wait": "decode_sums_kernel<no wait> + sums_offsets_kernel + decode_expand_kernel"}.get(dec.route, dec.route)
        tr = getattr(args, "traffic", None) or load_traffic(args.workload) or {}
        measured = bool(tr.get("measured"))
        out = {
            "metric": "compress+decompress GB/s (input bits), 1 GiB bitmap",
            "value": round(world * args.steps * in_bytes / elapsed / 1e9, 3),
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": desc, "words": n, "seed": args.seed, "bitmaps_per_gpu": 1,
                       "parallelism": f"column-shard x{world}, no collective"},
            "compression_ratio_C_over_N": round(c_words / n, 6),
            "compress_GBps": round(in_bytes / (comp_avg * 1e-3) / 1e9, 2),
            "decompress_GBps": round(in_bytes / (dec_avg * 1e-3) / 1e9, 2),
            "compress_ms": {"avg": round(comp_avg, 4), "min": round(comp_ms[0], 4), "median": round(comp_ms[len(comp_ms) // 2], 4)},
            "decompress_ms": {"avg": round(dec_avg, 4), "min": round(dec_ms[0], 4), "median": round(dec_ms[len(dec_ms) // 2], 4)},
            "copy_ceiling_GBps": round(copy_gbps, 1),
            "two_round_trips_in_flight_GBps": round(two_in_flight, 1) if two_in_flight else None,
            "roofline": {"kernel": "compress_pair_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": tr.get("compress_bytes_per_launch"), "traffic_measured_in_run": measured,
                         "traffic_source": tr.get("source"),
                         "algorithmic_bytes_per_launch": algo_c, "launch_ms": round(comp_avg, 4),
                         "launch_ms_isolated": round(comp_isolated, 4),
                         "frac_isolated": round(algo_c / (comp_isolated * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "frac_of_copy_ceiling": round(achieved / copy_gbps, 4)},
            "roofline_decompress": {"kernel": decode_kernels, "bound": "hbm",
                                    "achieved": round(achieved_d, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": round(achieved_d / HBM_PEAK_GBPS, 4),
                                    "traffic": tr.get("decompress_bytes_per_launch"), "traffic_measured_in_run": measured,
                                    "algorithmic_bytes_per_launch": algo_d, "launch_ms": round(dec_avg, 4)},
            "roofline_decompress_indexed": {"kernel": "decode_segments_kernel", "bound": "hbm",
                                            "achieved": round(algo_d / (idx_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                            "unit": "GB/s", "frac": round(algo_d / (idx_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                            "traffic": tr.get("decompress_indexed_bytes_per_launch"),
                                            "traffic_measured_in_run": measured,
                                            "algorithmic_bytes_per_launch": algo_d,
                                            "launch_ms": round(idx_ms, 4),
                                            "note": "side measurement with the segment index kept by the compressor; not part of value"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, param, args.cpu_sample_mib, args.seed)
        out.update(extra)
    # side block, never `value`: BASELINE configs[4] -- the 1024 independent 128 MiB columns, column c on rank c mod N --
    # so that the default command (the one the driver runs at N = 1, 2, 4, 8) also carries the column-shard figure
    if not args.no_columns:
        del comp, dec, d_in
        torch.cuda.empty_cache()
        side = columns_measurement(args, dev, rank, world, dist, False, args.columns_steps, 1, WORKLOADS["columns"][2])
        if rank == 0:
            out["columns"] = {k: side[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "scaling", "config",
                                                   "compression_ratio_C_over_N", "roofline")}
    if rank == 0:
        print(json.dumps(out), flush=True)
    finish()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None:
        if args.gpus > 1 or args.dry_launch:
            return launch_ranks(args, argv)
    elif int(world_env) != args.gpus:
        print(f"bench.py: started with WORLD_SIZE={world_env} but --gpus {args.gpus}: start it with matching values "
              f"(python bench.py --gpus N starts its own ranks)", file=sys.stderr)
        return 2
    # HBM bytes by the PMC counters, in child runs under rocprofv3, BEFORE this process touches the GPU (N = 1, real GPU only)
    args.traffic = None
    if args.gpus == 1 and world_env is None and not args.no_traffic and not os.environ.get("WAH_BENCH_REHEARSE") and args.workload != "columns":
        args.traffic = measure_traffic(args)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
