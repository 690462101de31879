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
