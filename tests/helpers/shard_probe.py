"""Rank body for tests/test_gpu_shard.py: SURVEY 8e on the HIP path without a multi-GPU node - WORLD_SIZE ranks, each
stepping its `make_sharded_env` slice of ONE seeded global batch on cuda:0, the episode metrics summed by
`gather_metrics` over a gloo group.  Every rank draws with the SAME seeds (device draws are functions of the global
env index, DIRAL_OPT_ENV_OFFSET).  Rank 0 prints one JSON line: the all-rank summary, and per rank the shard bounds and
checksums of its final state."""
import argparse
import json
import os

import torch
import torch.distributed as dist

from diral_amd.config import c2_config
from diral_amd.metrics import gather_metrics
from diral_amd.shard import make_sharded_env

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--slots", type=int, default=30)
ap.add_argument("--seed", type=int, default=1234)
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = c2_config(track_prr=True)
env, start = make_sharded_env(cfg, args.envs, device="cuda:0")
env.reset_topology(seed=args.seed)
for t in range(args.slots):
    env.step(env.sample(seed=1000 + t), t)
env.check()
summary = gather_metrics(env)
st = env.export_state(tables=True)
mine = {"rank": rank, "start": start, "count": env.B, "pos_x_sum": float(st["pos_x"].sum().item()),
        "seq_sum": int(st["seq"].to(torch.int64).sum().item()), "age_sum": int(st["age"].to(torch.int64).sum().item()),
        "kernel": env.last_kernel()}
allr = [None] * world
dist.all_gather_object(allr, mine)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"world": world, "summary": summary, "ranks": allr}), flush=True)
