"""The driver's random prefill (main_test.py:99-114) at C2, 4096 envs: the loop of one-slot calls against ONE launch of K
slots (diral_env_prefill).  python profiles/prefill_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diral_amd.config import STEP_DESIGN, c2_config
from diral_amd.driver import DriverLoop
from diral_amd.vec_env import VecV2VEnv

K, REP = 25, 12
RICH = dict(State=dict(add_channel_obs=True, add_reward=True, add_index=True, add_velocity=True, add_position=True))
for B, name, kw in ((64, "plain state (toy flags)", {}), (256, "plain state", {}), (1024, "plain state", {}), (4096, "plain state", {}),
                    (4096, "rich state (chobs + reward + index + velocity + position)", RICH)):
    cfg = c2_config(**kw)
    env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32)
    env.reset_topology(seed=1)
    loop = DriverLoop(env)
    loop.bootstrap(env.sample(0))
    for t in range(60):
        env.step(env.sample(t), t)
    torch.cuda.synchronize()

    def timed(fn):
        fn(0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(REP):
            fn(1000 * (r + 1))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (REP * K) * 1e6

    def loop_reference(seed):           # main_test.py:99-114 call by call: sample, my_step_design, obtain_state
        for k in range(K):
            a = env.sample(seed + k)
            obs, _ = env.my_step_design(a, 0)
            env.obtain_state(obs, a, loop._rews0)

    def loop_fused_step(seed):          # sample + ONE launch for my_step_design and obtain_state (own reward column)
        for k in range(K):
            env._step(STEP_DESIGN, env.sample(seed + k), 0)

    nxt = [env.sample(0)]
    def one_launch(seed):
        _, _, nxt[0] = env.prefill(nxt[0], K, seed, rew_in=loop._rews0)

    a, b_, c = timed(loop_reference), timed(loop_fused_step), timed(one_launch)
    print("B = %d, " % B + "%s\n  loop: sample + my_step_design + obtain_state   %7.2f us/slot\n  loop: sample + fused design step               %7.2f us/slot\n"
          "  diral_env_prefill, K = %d slots per launch      %7.2f us/slot   (%.0f %% less than the fused-step loop, %.0f %% less than the reference's call sequence; %.2f G agent-steps/s)"
          % (name, a, b_, K, c, 100 * (1 - c / b_), 100 * (1 - c / a), B * 64 / c / 1e3), flush=True)
