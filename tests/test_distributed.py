"""CPU tests of the N > 1 path: column sharding and bench.py's timing/aggregation harness, world size 2, gloo."""
import importlib
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_columns_partition():
    wah = importlib.import_module("gpu-wah_amd")
    for n, world in ((1024, 8), (1024, 1), (10, 4), (3, 8), (0, 2)):
        seen = []
        for r in range(world):
            mine = wah.columns.shard_columns(n, r, world)
            assert all(c % world == r for c in mine)
            seen += mine
        assert sorted(seen) == list(range(n))
    assert wah.columns.aggregate_throughput([10.0, 30.0], [1.0, 2.0]) == 20.0
    kinds = {wah.columns.column_spec(c, 992).kind for c in range(6)}
    assert kinds == {"sparse", "clustered", "dense"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_rehearsal():
    """Two processes, gloo, 127.0.0.1: disjoint columns, barrier-bracketed timing, max over ranks, one JSON line."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["world"] == 2
    assert res["owners"] == [1] * 6  # every column compressed by exactly one rank
    assert res["total_bytes"] == 4.0 * 992 * 64 * 6 * 2
    assert res["value"] > 0 and res["elapsed"] > 0
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]  # only rank 0 reports


def _run_bench(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_dry_launch_names_n_ranks():
    """`bench.py --gpus 8` without a launcher must start 8 ranks itself: the command it would run."""
    p = _run_bench(["--gpus", "8", "--steps", "3", "--warmup", "1", "--workload", "columns", "--dry-launch", "--master-port", "29777"])
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    cmd = res["dry_launch"]
    assert res["n_gpus"] == 8
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "3", "--warmup", "1", "--workload", "columns", "--master-port", "29777"]


def test_bench_refuses_mismatched_world_size():
    """Started by a launcher with a WORLD_SIZE other than --gpus: no line, non-zero exit (never a silent 1-GPU run)."""
    p = _run_bench(["--gpus", "8"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "WAH_BENCH_REHEARSE": "cpu"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_more_gpus_than_visible():
    """No GPU here: --gpus 2 must fail loudly, not fall back to fewer."""
    p = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_main_two_ranks_gloo():
    """bench.py's own main(): it starts two ranks (torch.distributed.run), they rendezvous over gloo on 127.0.0.1,
    shard the columns, time empty steps between barriers and rank 0 prints ONE line with n_gpus = 2."""
    for workload, extra in (("columns", ["--columns", "10"]), ("sparse", [])):
        p = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--master-port", str(_free_port())] + extra,
                       {"WAH_BENCH_REHEARSE": "cpu"})
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        res = json.loads(lines[0])
        assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1 and "rehearsal" in res
        assert res["scaling"] == ("strong" if workload == "columns" else "weak")
        if workload == "columns":
            assert res["config"]["columns"] == 10 and res["config"]["columns_per_gpu"] == 5


def test_bench_main_eight_ranks_gloo():
    """The world size the driver's scaling run ends with: bench.py starts EIGHT ranks itself (gloo, 127.0.0.1, empty steps,
    WAH_BENCH_REHEARSE=cpu), both workloads; the columns workload at BASELINE configs[4]'s shape: 1024 columns, 128 per
    rank, one launch of 128 columns per rank and step; ONE JSON line, n_gpus = 8."""
    for workload in ("columns", "sparse"):
        p = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--workload", workload, "--master-port", str(_free_port())],
                       {"WAH_BENCH_REHEARSE": "cpu", "OMP_NUM_THREADS": "1"}, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        res = json.loads(lines[0])
        assert res["n_gpus"] == 8 and res["steps"] == 2 and res["warmup"] == 1 and "rehearsal" in res
        assert res["scaling"] == ("strong" if workload == "columns" else "weak")
        if workload == "columns":
            assert res["config"]["columns"] == 1024 and res["config"]["columns_per_gpu"] == 128


def test_bench_column_plan_covers_every_column_once():
    import bench

    for n_columns, world, per_launch in ((1024, 8, 128), (1024, 1, 128), (10, 4, 3), (3, 8, 128)):
        seen = []
        for r in range(world):
            mine, batches = bench.column_plan(n_columns, r, world, per_launch)
            assert [c for b in batches for c in b] == mine and all(len(b) <= per_launch for b in batches)
            seen += mine
        assert sorted(seen) == list(range(n_columns))
