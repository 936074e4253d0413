"""Rank body for tests/test_spawn.py: gloo group over 127.0.0.1, one all-reduce, rank 0 prints one JSON line.
`--fail-rank R` makes rank R exit with status 3 before the collective (rc propagation)."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--fail-rank", type=int, default=-1)
ap.add_argument("--gpus", type=int, default=1)
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == args.fail_rank:
    sys.exit(3)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"world": world, "sum": float(t.item()), "master": os.environ["MASTER_ADDR"],
                      "port": int(os.environ["MASTER_PORT"]), "gpus": args.gpus}), flush=True)
