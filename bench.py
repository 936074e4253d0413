#!/usr/bin/env python3
"""bench.py - headline benchmark: agent-steps/sec of the fused V2V env step.

Metric (BASELINE.json): agent-steps/sec (envs x vehicles) on the 64-UE/32-res
batched env.  One "step" = one time-slot of ALL B envs of this rank = one launch
of the fused kernel (stamp -> collide -> reward -> closest-tx -> gossip merge ->
move -> observe), actions pre-generated in HBM, outputs written to HBM.

  python bench.py --gpus 1 --steps 200 --warmup 50
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; envs are independent so ranks share nothing on the step
path (weak scaling: B per GPU fixed); RCCL is used once, to all-reduce the
episode metrics.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from diral_amd.config import bench_config  # noqa: E402
from diral_amd.metrics import gather_metrics  # noqa: E402
from diral_amd.roofline import HBM_PEAK_GBPS, algorithmic_bytes_per_env_slot  # noqa: E402
from diral_amd.vec_env import VecV2VEnv  # noqa: E402

WORKLOADS = {
    # name: (N, A, L, B per GPU, mobility_vary)  - SURVEY.md section 8 table
    "c2": (64, 32, 2000.0, 4096, False),    # BASELINE.json configs[1]: the metric's config
    "c3": (256, 64, 4000.0, 8192, False),   # configs[2] congested
    "c5": (128, 64, 4000.0, 16384, True),   # configs[4] dynamic density
}


def usable_cores() -> int:
    """Host threads this process may really use: sched affinity capped by the
    cgroup CPU quota (a 256-thread box with a 16-CPU quota has 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, seconds_target: float = 12.0):
    """Time the CPU oracle (oracle/diral_oracle.c, the reference restated in C,
    reference-faithful pow() mode) on this host's cores on a bounded sample of
    the same workload.  A reported baseline, not the target."""
    import numpy as np
    from oracle.oracle import Oracle, SQ_POW, has_openmp
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    cores = usable_cores()
    threads = min(cores, 64) if has_openmp() else 1
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    B = max(threads * 16, 64)
    rng = np.random.default_rng(1234)
    o = Oracle(cfg, batch=B, sq_mode=SQ_POW, threads=threads)
    o.reset(rng.integers(0, int(L), size=(B, N)).astype(np.float64), np.zeros((B, N)),
            rng.uniform(1.1, 2.7, size=(B, N)))
    warm, slots = 25, 0
    acts = rng.integers(0, A, size=(64, B, N)).astype(np.int32)
    for t in range(warm):
        r, c = o.step(0, acts[t % 64], t)
        o.obtain_state(acts[t % 64], c, r)
    t0 = time.perf_counter()
    t = warm
    while True:
        r, c = o.step(0, acts[t % 64], t)
        o.obtain_state(acts[t % 64], c, r)
        t += 1
        slots += 1
        el = time.perf_counter() - t0
        if el >= seconds_target or slots >= 20000:
            break
    return {
        "value": B * N * slots / el, "unit": "agent-steps/s", "cores": threads, "kind": "port",
        "sample": "%d envs x %d slots of the same %d-UE/%d-res workload after %d warm-up slots, "
                  "oracle/diral_oracle.c (C restatement pinned to the reference goldens), "
                  "OpenMP over envs on %d host threads, %.1f s" % (B, slots, N, A, warm, threads, el),
    }


def prr_parity(cfg, device, envs: int = 64, slots: int = 60):
    """PRR parity on a bounded sample (BASELINE.json: 'PRR matching the reference
    within 1e-6'): the same seeded envs and actions through the HIP path
    (my_step_ch, test_env.py:384-405 ratio R per transmission) and through the CPU
    oracle in its reference-faithful mode; episode PRR = mean R over transmissions."""
    import numpy as np
    from oracle.oracle import Oracle, SQ_POW
    from diral_amd.config import STEP_MY_STEP_CH
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    rng = np.random.default_rng(4321)
    x0 = rng.integers(0, int(L), size=(envs, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(envs, N))
    env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float64, step_mode="my_step_ch")
    env.reset_topology(x0, None, v0)
    orc = Oracle(cfg, batch=envs, sq_mode=SQ_POW, threads=min(usable_cores(), 16))
    orc.reset(x0, np.zeros((envs, N)), v0)
    for t in range(slots):
        a = rng.integers(0, A, size=(envs, N)).astype(np.int32)
        env.step(a, t)
        orc.step(STEP_MY_STEP_CH, a, t)
    m, om = env.metrics().cpu().numpy(), orc.metrics()
    gpu = float(m[:, 4].sum() / m[:, 5].sum())
    cpu = float(om[:, 4].sum() / om[:, 5].sum())
    per_env = np.abs(m[:, 4] / m[:, 5] - om[:, 4] / om[:, 5]).max()
    return {"gpu": gpu, "cpu_oracle": cpu, "abs_diff": abs(gpu - cpu), "max_abs_diff_per_env": float(per_env),
            "tolerance": 1e-6, "sample": "%d envs x %d slots, my_step_ch, reward_design %d" % (envs, slots, cfg.reward_design)}


def kernel_name(N: int, A: int, out_dtype: str, step_mode: str = "my_step") -> str:
    """The step kernel csrc/diral_env.hip dispatches for the bench configuration
    (default State flags, my_step / my_step_ch, all y == 0)."""
    o64 = "true" if out_dtype == "f64" else "false"
    ch = "true" if step_mode == "my_step_ch" else "false"
    if N <= 64 and A <= 64:
        return "diral::step_fast64_kernel<true,%s,%s,false>" % (o64, ch)
    if 64 < N <= 256 and A <= 64:
        return "diral::step_wide_kernel<%d,%s,%s,%s,false>" % (2 if N <= 128 else 4, o64,
                                                                "true" if N in (128, 256) else "false", ch)
    return "diral::step_kernel<%d,%s>" % (1 if N <= 64 else 2 if N <= 128 else 4,
                                           "true" if (out_dtype == "f32" and ch == "false") else "false")


def load_traffic(workload: str):
    """HBM bytes per launch from the committed PMC summary (profiles/), if any."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        return d.get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="envs per GPU (default: the workload's)")
    ap.add_argument("--out-dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--sticky", type=float, default=0.0,
                    help="probability an agent keeps its resource (0 = iid uniform, worst case)")
    ap.add_argument("--step-mode", default="my_step", choices=["my_step", "my_step_ch"],
                    help="my_step = the metric's step kind; my_step_ch = the PRR-reward variant (test_env.py:351-443)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus),
                  file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the RCCL group is always created - also for
    # one rank, so the single-GPU run exercises the same collective code path
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    N, A, L, B, vary = WORKLOADS[args.workload]
    if args.batch > 0:
        B = args.batch
    cfg = bench_config(N, A, L, mobility_vary=vary)
    out_dtype = torch.float32 if args.out_dtype == "f32" else torch.float64
    env = VecV2VEnv(cfg, batch=B, device=device, out_dtype=out_dtype, step_mode=args.step_mode)
    env.reset_topology(seed=1234 + rank)

    # synthetic actions, resident in HBM before the timed region: a ring of
    # pre-drawn [B,N] tensors (iid uniform, or sticky to mimic a converged policy)
    ring = 32
    acts = [env.sample(seed=1000 * rank + i) for i in range(ring)]
    if args.sticky > 0:
        g = torch.Generator(device=device).manual_seed(99 + rank)
        for i in range(1, ring):
            keep = torch.rand((B, N), device=device, generator=g) < args.sticky
            acts[i] = torch.where(keep, acts[i - 1], acts[i])

    ei = cfg.episode_interval
    t = 0

    def one_step(t):
        env.step(acts[t % ring], t)
        if t % ei == ei - 1:
            env.update_velocity(seed=t)      # main_test.py:226-233 (no-op unless mobility_vary)

    for _ in range(args.warmup):
        one_step(t)
        t += 1
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize(device)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        one_step(t)
        t += 1
    ev1.record()
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize(device)
    wall = time.perf_counter() - t_start
    kernel_ms = ev0.elapsed_time(ev1) / args.steps      # HIP events on the launch stream

    env.check()
    totals = gather_metrics(env)                         # RCCL all-reduce (metrics only)
    if use_dist:
        w = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())

    if rank == 0:
        agent_steps = float(B) * N * args.steps * world
        bytes_launch = algorithmic_bytes_per_env_slot(N, A, cfg.state_space) * B
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "agent-steps/sec (envs x vehicles), %d-UE/%d-res batched env" % (N, A),
            "value": agent_steps / wall,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d-UE/%d-res, batch=%d envs per GPU x %d GPU, L=%g m, Rc=%g, K=%d bins, "
                            "reward_design=2, %s+obtain_state fused, %s actions, %s outputs" % (
                                args.workload, N, A, B, world, L, cfg.communication_range,
                                cfg.State.num_bins, args.step_mode, "iid-uniform" if args.sticky == 0 else
                                "sticky(p=%.2f)" % args.sticky, args.out_dtype),
                "batch_per_gpu": B, "num_users": N, "num_channels": A, "state_space": cfg.state_space,
                "parallelism": "env-shard x%d (no data-path collective)" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": load_traffic(args.workload),
                "kernel": kernel_name(N, A, args.out_dtype, args.step_mode),
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": bytes_launch,
            },
            "episode_metrics": totals,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
            line["prr_parity"] = prr_parity(cfg, device)
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
