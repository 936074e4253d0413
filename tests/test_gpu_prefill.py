"""The driver's random prefill (main_test.py:99-114: sample -> my_step_design -> obtain_state, every state kept) as ONE
launch of K slots (`diral_env_prefill`, step_fast64_slots_kernel with PolParams::prefill) against the loop of one-slot
calls it replaces: states, actions, tables, positions, metrics - bit for bit."""
import numpy as np
import pytest
import torch

from diral_amd.config import (ERR_UNSUPPORTED, KERNEL_FAST64, KERNEL_POLICY, bench_config, c2_config)
from diral_amd.driver import DriverLoop
from diral_amd.vec_env import DiralError, VecV2VEnv

pytestmark = pytest.mark.gpu

RICH = dict(add_channel_obs=True, add_reward=True, add_index=True, add_velocity=True, add_position=True)


def _pair(cfg, B, dtype, seed):
    envs = []
    rng = np.random.default_rng(seed)
    x0 = rng.integers(0, int(cfg.highway_length), size=(B, cfg.num_users)).astype(np.float64)
    v0 = np.full((B, cfg.num_users), 1.7) if cfg.mobility_vary else rng.uniform(1.1, 2.7, size=(B, cfg.num_users))
    for _ in range(2):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=dtype)
        env.reset_topology(x0, 0.0, v0)
        envs.append(env)
    return envs


@pytest.mark.parametrize("cfg,K,dtype", [
    (c2_config(), 7, torch.float32),
    (c2_config(), 30, torch.float64),
    (c2_config(State=RICH, enable_fingerprint=True), 12, torch.float64),
    (c2_config(State=RICH, reward_design=1), 1, torch.float32),
    (bench_config(40, 12, 900.0, State=RICH, mobility_vary=True), 26, torch.float32),        # dense: most collisions inside 2 Rc
    (bench_config(64, 8, 9000.0, communication_range=100.0), 9, torch.float32),             # sparse: keyed quads
])
def test_prefill_in_one_launch_equals_the_loop_of_one_slot_calls(cfg, K, dtype):
    B, seed = 24, 77001
    e_loop, e_one = _pair(cfg, B, dtype, 5)
    loops = [DriverLoop(e) for e in (e_loop, e_one)]
    a0 = e_loop.sample(123)
    for lp in loops:
        lp.bootstrap(a0)                                             # my_step: `rews` of main_test.py:92, the stale reward column
    want_s, want_a = [], []
    for k in range(K):
        a = e_loop.sample(seed + k)
        want_a.append(a.clone())
        want_s.append(loops[0].prefill_step(a))
    states, acts, nxt = e_one.prefill(e_one.sample(seed), K, seed, rew_in=loops[1]._rews0)
    torch.cuda.synchronize()
    assert (e_one.last_kernel() & 15) == KERNEL_FAST64 and (e_one.last_kernel() & KERNEL_POLICY)
    assert torch.equal(acts, torch.stack(want_a)) and torch.equal(nxt, e_loop.sample(seed + K))
    for k in range(K):
        assert torch.equal(states[k], want_s[k]), (k, (states[k] != want_s[k]).nonzero()[:5])
    s1, s2 = e_loop.export_state(), e_one.export_state()
    for key in s1:
        assert torch.equal(s1[key], s2[key]), key
    assert torch.equal(e_loop.metrics(), e_one.metrics())
    # ... and the envs go on alike (what the launch wrote back: tables, ring, positions)
    for t in range(3):
        a = e_loop.sample(900 + t)
        o1, r1, _ = e_loop.step(a, t)
        o2, r2, _ = e_one.step(a, t)
        assert torch.equal(o1, o2) and torch.equal(r1, r2)
    e_loop.check(); e_one.check()


def test_driver_loop_prefill_takes_the_launch_where_it_can_and_loops_elsewhere():
    for cfg, fused in ((c2_config(), True), (bench_config(128, 16, 4000.0), False),
                       (c2_config(track_arrival=True), False)):
        e1, e2 = _pair(cfg, 6, torch.float64, 9)
        l1, l2 = DriverLoop(e1), DriverLoop(e2)
        a0 = e1.sample(1)
        l1.bootstrap(a0); l2.bootstrap(a0)
        states, acts = l2.prefill(5, 4400)
        assert bool(e2.last_kernel() & KERNEL_POLICY) == fused
        for k in range(5):
            a = e1.sample(4400 + k)
            assert torch.equal(acts[k], a) and torch.equal(states[k], l1.prefill_step(a))
        if not fused:
            with pytest.raises(DiralError) as ei:
                e2.prefill(a0, 3, 1)
            assert ei.value.status == ERR_UNSUPPORTED
