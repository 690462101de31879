"""One rank of the world-size-2 CPU rehearsal of bench.py's multi-GPU path (gloo backend).

The timed body is a stand-in (the CPU oracle compressing this rank's columns) because there is no GPU here;
everything around it -- rendezvous, column sharding, barrier, max-over-ranks timing, whole-job aggregation,
the single JSON line from rank 0 -- is the code bench.py runs on the GPUs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib

    wah = importlib.import_module("gpu-wah_amd")
    from tests import _oracle

    oracle = _oracle.load()
    n_columns, n_words = 6, 992 * 64
    mine = wah.columns.shard_columns(n_columns, rank, world)
    specs = [wah.columns.column_spec(c, n_words) for c in mine]
    data = [oracle.gen_uniform(s.n_words, s.seed, 0.01) if s.kind == "sparse" else
            oracle.gen_uniform(s.n_words, s.seed, 0.5) if s.kind == "dense" else
            oracle.gen_clustered(s.n_words, s.seed) for s in specs]
    sizes = []

    def step():
        sizes.clear()
        for d in data:
            sizes.append(len(oracle.compress(d)))

    elapsed = bench.timed_steps(step, steps=2, warmup=1, dist=dist, device=None)
    bytes_mine = 4.0 * n_words * len(mine) * 2
    total = torch.tensor([bytes_mine], dtype=torch.float64)
    dist.all_reduce(total)
    owned = [torch.zeros(n_columns, dtype=torch.int64) for _ in range(world)]
    mask = torch.zeros(n_columns, dtype=torch.int64)
    mask[mine] = 1
    dist.all_gather(owned, mask)
    if rank == 0:
        print(json.dumps({"elapsed": elapsed, "total_bytes": float(total.item()), "world": world,
                          "owners": [int(sum(o[c] for o in owned)) for c in range(n_columns)],
                          "value": float(total.item()) / elapsed}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
