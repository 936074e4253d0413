#!/usr/bin/env python3
"""End-to-end example: the reference driver's slot loop (main_test.py:86-236) on
B parallel envs, with the SPS baseline (algorithms/v2x_sps.py) as the policy.
Everything stays on the GPU; the only host traffic is the final metric read-out.

  python examples/rollout_sps.py --envs 1024 --slots 500
  python -m torch.distributed.run --nproc-per-node 8 examples/rollout_sps.py --envs 262144   # sharded
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from diral_amd import c2_config  # noqa: E402
from diral_amd.driver import DriverLoop  # noqa: E402
from diral_amd.metrics import gather_metrics  # noqa: E402
from diral_amd.shard import make_sharded_env, rank_world  # noqa: E402
from diral_amd.sps import SpsPolicy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024, help="total envs over all ranks")
    ap.add_argument("--slots", type=int, default=500)
    ap.add_argument("--policy", choices=["sps", "random"], default="sps")
    ap.add_argument("--fused", action="store_true", help="one launch per slot (random policy only: SPS needs chobs)")
    ap.add_argument("--one-launch", action="store_true",
                    help="SPS policy: the whole slot (env step + reward shaping + SPS decision) as ONE launch, "
                         "diral_env_step_policy - the channel observation never leaves the chip")
    ap.add_argument("--slots-per-launch", type=int, default=1,
                    help="with --one-launch: K slots in ONE launch (DiralSlotPolicy::slots): the env stays on the chip from "
                         "slot to slot, only the per-slot shaped rewards and the metrics leave it")
    args = ap.parse_args()
    rank, local_rank, world = rank_world()
    torch.cuda.set_device(local_rank)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cfg = c2_config()
    # io_ring=2: state and next_state of consecutive slots live in two alternating output sets (no copies)
    env, start = make_sharded_env(cfg, args.envs, out_dtype=torch.float32, io_ring=2)
    env.reset_topology(seed=1234)                     # one global seed: the shard offset selects the envs
    loop = DriverLoop(env, global_reward_avg=True, episode_interval=cfg.episode_interval,
                      fused=args.fused and args.policy == "random")
    pol = SpsPolicy(env.B, env.N, env.A, device=env.device, seed=start)
    state = loop.bootstrap(pol.prev_action)
    actions = pol.prev_action.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.one_launch and args.policy == "sps":
        acts = [actions, torch.empty_like(actions)]
        K = max(1, args.slots_per_launch)
        if K > 1:
            args.slots -= args.slots % K
            shaped = torch.empty((K, env.B, env.N), dtype=torch.float32, device=env.device)
            for n in range(args.slots // K):
                # (want_obs=False: a policy-only rollout - no state vector, so no histogram either)
                env.step_policy(acts[n & 1], n * K, pol, acts[(n + 1) & 1], shaped_out=shaped, global_reward_avg=True,
                                slots=K, want_obs=False)
                if n == args.slots // K // 2:
                    env.metrics(clear=True)
        else:
            shaped = torch.empty((env.B, env.N), dtype=torch.float32, device=env.device)
            for t in range(args.slots):
                state, _, _ = env.step_policy(acts[t & 1], t, pol, acts[(t + 1) & 1], shaped_out=shaped, global_reward_avg=True)
                if t == args.slots // 2:
                    env.metrics(clear=True)
        args.slots_done = True
    for t in range(0 if not getattr(args, "slots_done", False) else args.slots, args.slots):
        out = loop.slot(actions, t)                   # one fused env launch + reward shaping
        state = out["next_state"]                     # what a learning agent would consume
        if args.policy == "sps":
            actions = pol.step_from_chobs(env._chobs, actions)      # window + SPS decision in one launch
        else:
            actions = env.sample(seed=t)
        if out["episode_end"]:
            loop.end_episode()
        if t == args.slots // 2:
            env.metrics(clear=True)                   # report the second half only
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = gather_metrics(env)
    if rank == 0:
        print("policy=%s envs=%d slots=%d  %.3g agent-steps/s (env + policy + shaping)" % (
            args.policy, args.envs, args.slots, args.envs * env.N * args.slots / dt))
        print("collision fraction %.3f  mean reward %.3f  state %s" % (
            m["collision_fraction"], m["mean_reward_per_agent_step"], tuple(state.shape)))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
