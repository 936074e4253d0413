"""Episode-metric gather across ranks.

Envs are independent (one ``TestEnv`` per process in the reference,
main_test.py:46), so the step path has NO collective.  RCCL over xGMI is used
only here: one all-reduce of six float64 sums per report (latency-bound, link
bandwidth irrelevant).  Works unchanged on the gloo backend (CPU tests).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from .config import M_COLUMNS, M_PRR_CNT, M_PRR_SUM, M_SLOTS, M_SUM_REWARD, M_TX_COLLIDED, M_TX_SOLE


def reduce_metric_sums(local: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """local: [B_local, M] per-env accumulators -> [M] sums over every env of
    every rank (all-reduce SUM when torch.distributed is initialised)."""
    if local.dim() != 2 or local.shape[1] != M_COLUMNS:
        raise ValueError("expected [B, %d] metrics, got %s" % (M_COLUMNS, tuple(local.shape)))
    sums = local.to(torch.float64).sum(dim=0)
    # env count rides along so means can be formed without a second collective
    packed = torch.cat([sums, torch.tensor([float(local.shape[0])], dtype=torch.float64, device=sums.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if packed.is_cuda and dist.get_backend(group) == "gloo":
            # (a gloo group - CPU tests, or ranks that share one GPU: 7 doubles through host memory)
            host = packed.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            packed = host.to(packed.device)
        else:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


def summarize(packed: torch.Tensor, num_channels: Optional[int] = None) -> Dict[str, float]:
    p = packed.detach().cpu().tolist()
    sums, n_env = p[:M_COLUMNS], p[M_COLUMNS]
    slots = sums[M_SLOTS]
    tx = sums[M_TX_SOLE] + sums[M_TX_COLLIDED]
    out = {
        "envs": n_env,
        "env_slots": slots,
        "sum_reward": sums[M_SUM_REWARD],
        "mean_reward_per_agent_step": sums[M_SUM_REWARD] / tx if tx else 0.0,
        "collision_fraction": sums[M_TX_COLLIDED] / tx if tx else 0.0,
        "prr": sums[M_PRR_SUM] / sums[M_PRR_CNT] if sums[M_PRR_CNT] else None,
    }
    if num_channels and slots:
        # the driver's per-slot `collision = num_channels - sum_r` (main_test.py:178), averaged
        out["mean_collision_metric"] = num_channels - sums[M_SUM_REWARD] / slots
    return out


def gather_metrics(env, group: Optional[dist.ProcessGroup] = None, clear: bool = False) -> Dict[str, float]:
    """All-rank episode metrics of a VecV2VEnv."""
    local = env.metrics(clear=clear)
    return summarize(reduce_metric_sums(local, group), env.A)
