"""Env sharding across the GPUs of a node.

The reference runs one env per process (main_test.py:46); envs never interact,
so a batch of B_total envs shards on the batch axis with NO data-path
collective: rank r owns envs [start, start+count).  RCCL is only used by
:mod:`diral_amd.metrics` to sum the episode metrics.
"""
from __future__ import annotations

import os
from typing import Tuple


def env_shard(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """(start, count) of the contiguous env range of `rank`; the first
    ``total_envs % world`` ranks get one extra env."""
    if world < 1 or not (0 <= rank < world) or total_envs < 0:
        raise ValueError("bad shard request: total=%d rank=%d world=%d" % (total_envs, rank, world))
    base, rem = divmod(total_envs, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def make_sharded_env(cfg, total_envs: int, device=None, **kw):
    """This rank's VecV2VEnv slice of a `total_envs` batch (e.g. BASELINE
    configs[3]: 262144 envs over 8 GPUs = 32768 per GPU).  Returns (env, start).
    `device`: default cuda:LOCAL_RANK (one process per GPU); a box with fewer GPUs than ranks names one
    (tests/test_gpu_shard.py runs two ranks on cuda:0)."""
    import torch
    from .vec_env import VecV2VEnv
    rank, local_rank, world = rank_world()
    start, count = env_shard(total_envs, rank, world)
    # the shard's global env offset: device draws (topology, sample, velocity) are a function of
    # (seed, global env index), so every rank can use the SAME seed and the job draws what one
    # handle holding all `total_envs` envs would draw
    kw.setdefault("env_offset", start)
    env = VecV2VEnv(cfg, batch=count, device=torch.device("cuda", local_rank) if device is None else device, **kw)
    return env, start
