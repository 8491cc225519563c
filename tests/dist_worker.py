"""Worker of tests/test_bench_dist.py: the N>1 bookkeeping of bench.py on CPU with gloo."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu")
    dist.barrier()
    mine = 1.0 + 0.5 * rank  # this rank's "measured" seconds
    whole = bench.max_over_ranks(mine, dev)
    assert abs(whole - (1.0 + 0.5 * (world - 1))) < 1e-12, whole
    value = bench.job_throughput(world, 3, 20, whole)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "elapsed": whole, "value": value}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
