"""Randomised configuration sweep: every State flag / reward design / step kind /
size combination the build accepts, drawn from a fixed seed, run for a few dozen slots
on the HIP path and on the oracle - bit-exact (exp() rewards within the documented
absolute tolerance).  Exercises the general kernel's flag handling and the dispatch to
the specialised kernels with configurations nobody wrote a dedicated test for."""
import numpy as np
import pytest

from diral_amd.config import (ConfigError, EnvConfig, KERNEL_FAST64, KERNEL_GENERAL, KERNEL_WIDE, STEP_DESIGN,
                              STEP_MY_STEP, STEP_MY_STEP_CH, bench_config)
from tests.test_gpu_parity import random_rollout

pytestmark = pytest.mark.gpu


def draw_case(i):
    rng = np.random.default_rng(9000 + i)
    N = int(rng.choice([2, 3, 5, 17, 31, 64, 65, 90, 128, 150, 256]))
    A = int(rng.choice([1, 2, 3, 7, 16, 32, 40, 64]))
    K = int(rng.choice([1, 3, 10, 20, 21, 40, 64]))
    mode = int(rng.choice([STEP_MY_STEP, STEP_MY_STEP, STEP_MY_STEP_CH, STEP_DESIGN]))
    rd = int(rng.choice([2, 3, 4])) if mode == STEP_MY_STEP_CH else int(rng.integers(1, 6))
    L = float(rng.choice([8.0, 20.0, 40.0]) * N + rng.integers(20, 200))
    state = dict(type=int(rng.choice([1, 2])), add_reward=bool(rng.random() < 0.3), add_action=bool(rng.random() < 0.8),
                 add_index=bool(rng.random() < 0.3), add_velocity=bool(rng.random() < 0.3),
                 action_index=str(rng.choice(["binary", "real"])), add_position=bool(rng.random() < 0.3),
                 add_positional_dist=bool(rng.random() < 0.2), add_positional_dist_piggy=bool(rng.random() < 0.8),
                 add_positional_dist_type=int(rng.choice([1, 2, 2])), num_bins=K,
                 add_channel_obs=bool(rng.random() < 0.4))
    if state["type"] == 1 and state["add_positional_dist_piggy"]:
        state["type"] = 2          # the reference crashes on this pair when no tx is in range
    cfg = bench_config(N, A, L, reward_design=rd, State=state,
                       mobility=bool(rng.random() < 0.85), mobility_vary=bool(rng.random() < 0.4),
                       enable_fingerprint=bool(rng.random() < 0.3),
                       proportional_fair=bool(rng.random() < 0.2 and mode == STEP_MY_STEP),
                       congestion_test=bool(rng.random() < 0.2),
                       communication_range=float(rng.choice([30.0, 120.0, 250.0])),
                       bin_range=float(rng.choice([100.0, 500.0])))
    return cfg, mode, float(rng.choice([0.0, 0.5, 0.9])), int(rng.choice([0, 7]))


@pytest.mark.parametrize("i", range(96))
def test_random_configuration_vs_oracle(i):
    cfg, mode, sticky, vel_every = draw_case(i)
    if not (cfg.mobility or cfg.enable_design_topology):
        # the ONLY rejection this generator can draw: no vehicles are built without mobility or the
        # design topology (network.py:54-60).  Anything else validate() raises is a regression.
        with pytest.raises(ConfigError, match="mobility"):
            cfg.validate()
        # ... with the design topology the same draw is a STATIC topology (network.py:302-305: update_mobility
        # does nothing): a run-time switch of the specialised kernels' EXTRA instantiations
        cfg = cfg.replace(enable_design_topology=True)
    cfg.validate()
    B = 4 if cfg.num_users > 64 else 12
    # the general kernel (DIRAL_OPT_KERNEL_PATH), PRR metrics tracked in my_step as well ...
    random_rollout(cfg, B=B, T=26, seed=100 + i, mode=mode, sticky=sticky, vel_every=vel_every or None, threads=8,
                   track_prr=True, force_general=True, expect_kernel=KERNEL_GENERAL)
    # ... and whatever the dispatch picks: step_fast64 / step_wide for EVERY State block at A <= 64 on the one-lane
    # highway - with or without piggybacked tables (no tables: the table-less step), mobile or static, PRR metrics
    # or not (the secondary observation modes get their columns from posdist_kernel afterwards)
    special = cfg.num_channels <= 64
    want = (KERNEL_FAST64 if cfg.num_users <= 64 else KERNEL_WIDE) if special else KERNEL_GENERAL
    random_rollout(cfg, B=B, T=26, seed=100 + i, mode=mode, sticky=sticky, vel_every=vel_every or None, threads=8,
                   track_prr=bool(i & 1), expect_kernel=want)


def draw_fast_case(i):
    rng = np.random.default_rng(7000 + i)
    N = int(rng.integers(1, 257))
    A = int(rng.integers(1, 65))
    K = int(rng.integers(1, 65))
    mode = int(rng.choice([STEP_MY_STEP, STEP_MY_STEP_CH, STEP_DESIGN]))
    rd = int(rng.choice([2, 3, 4])) if mode == STEP_MY_STEP_CH else int(rng.integers(1, 6))
    L = float(rng.choice([6.0, 15.0, 40.0]) * N + rng.integers(20, 300))
    cfg = bench_config(N, A, L, reward_design=rd, State=dict(num_bins=K), mobility_vary=bool(rng.random() < 0.5),
                       congestion_test=bool(rng.random() < 0.15), communication_range=float(rng.choice([40.0, 150.0, 250.0])),
                       bin_range=float(rng.choice([200.0, 500.0])))
    return cfg, mode, float(rng.choice([0.0, 0.7])), bool(rng.random() < 0.5)


@pytest.mark.parametrize("i", range(64))
def test_random_default_flag_configuration_on_the_specialised_kernels(i):
    """Default State flags (what step_fast64 / step_wide serve), random sizes, reward
    designs, step kinds, densities and output dtypes: `env.step` (no channel-obs output,
    so the specialised kernels run) against the oracle, bit for bit."""
    import torch
    from oracle.oracle import Oracle, SQ_IEEE
    from tests.test_gpu_parity import EXP_ATOL, make_env
    cfg, mode, sticky, f64 = draw_fast_case(i)
    cfg.validate()
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    B = 3 if N > 64 else 10
    rng = np.random.default_rng(500 + i)
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    v0 = np.full((B, N), 1.7) if cfg.mobility_vary else rng.uniform(1.1, 2.7, size=(B, N))
    env = make_env(cfg, B, mode={STEP_MY_STEP: "my_step", STEP_MY_STEP_CH: "my_step_ch", STEP_DESIGN: "my_step_design"}[mode],
                   dtype=torch.float64 if f64 else torch.float32)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    env.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
    exp_rew = cfg.reward_design in (3, 4) and mode != STEP_DESIGN
    for t in range(32):
        new = rng.integers(0, A, size=(B, N))
        acts = np.where(rng.random((B, N)) < sticky, acts, new).astype(np.int32)
        obs, rew, _ = env.step(acts, t)
        o_rew, o_chobs = orc.step(mode, acts, t)
        o_state = orc.obtain_state(acts, o_chobs, o_rew)
        torch.cuda.synchronize()
        assert np.array_equal(obs.cpu().numpy(), o_state if f64 else o_state.astype(np.float32)), t
        if exp_rew:
            assert np.all(np.abs(rew.double().cpu().numpy() - o_rew) <= (EXP_ATOL if f64 else 1e-6)), t
        else:
            assert np.array_equal(rew.cpu().numpy(), o_rew if f64 else o_rew.astype(np.float32)), t
        if t % 9 == 8:
            d = rng.integers(1, 4, size=(B, N)).astype(np.uint8)
            env.update_velocity(d)
            orc.update_velocity(d)
    st, oe = env.export_state(), orc.export()
    assert np.array_equal(st["seq"].cpu().numpy(), oe["seq"])
    assert np.array_equal(st["x"].cpu().numpy(), oe["x"])
    assert np.array_equal(st["age"].cpu().numpy(), np.minimum(oe["age"], 255))
    assert np.array_equal(st["pos_x"].cpu().numpy(), oe["pos_x"])
    env.check()


def draw_rich_case(i):
    rng = np.random.default_rng(11000 + i)
    N = int(rng.choice([1, 2, 7, 33, 63, 64, 65, 100, 128, 129, 200, 256]))
    A = int(rng.choice([1, 3, 8, 31, 32, 33, 64]))
    K = int(rng.choice([1, 2, 9, 20, 40, 64]))
    if N > 128 and K > 40:
        K = 40              # N > 128 with 64 bins exceeds the general kernel's 160 KB LDS budget (DIRAL_ERR_UNSUPPORTED)
    mode = int(rng.choice([STEP_MY_STEP, STEP_MY_STEP, STEP_MY_STEP_CH, STEP_DESIGN]))
    rd = int(rng.choice([2, 3, 4])) if mode == STEP_MY_STEP_CH else int(rng.integers(1, 6))
    L = float(rng.choice([6.0, 15.0, 40.0]) * N + rng.integers(20, 300))
    state = dict(type=int(rng.choice([1, 2])), add_reward=bool(rng.random() < 0.5), add_action=bool(rng.random() < 0.8),
                 add_index=bool(rng.random() < 0.5), add_velocity=bool(rng.random() < 0.5),
                 action_index=str(rng.choice(["binary", "real"])), add_position=bool(rng.random() < 0.5),
                 add_channel_obs=bool(rng.random() < 0.6), num_bins=K)
    cfg = bench_config(N, A, L, reward_design=rd, State=state, mobility_vary=bool(rng.random() < 0.5),
                       enable_fingerprint=bool(rng.random() < 0.5), congestion_test=bool(rng.random() < 0.15),
                       communication_range=float(rng.choice([40.0, 150.0, 250.0])),
                       bin_range=float(rng.choice([200.0, 500.0])),
                       track_arrival=bool(rng.random() < 0.3))
    return cfg, mode, float(rng.choice([0.0, 0.7])), bool(rng.random() < 0.5), bool(N <= 64 and rng.random() < 0.3)


@pytest.mark.parametrize("i", range(72))
def test_random_rich_configuration_on_the_specialised_kernels(i):
    """The RICH instantiations (csrc/rich_out.hpp): random cheap State flags (test_env.py:527-583),
    State.type 1 / 2, every step kind, both output dtypes, vehicles on or off the y = 0 lane,
    arrival stamps on or off - through BOTH call patterns: the fused `_step` with the channel
    observation requested, and the reference's `obs, rews = env.my_step*(a, t)` +
    `env.obtain_state(obs, a, rews, episode, eps)` (main_test.py:144-164), which must be served by
    the same single launch.  State, channel observation and rewards against the oracle, bit for
    bit (exp() rewards within the documented tolerance)."""
    import torch
    from oracle.oracle import Oracle, SQ_IEEE
    from tests.test_gpu_parity import EXP_ATOL, make_env
    cfg, mode, sticky, f64, offlane = draw_rich_case(i)
    cfg.validate()
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    B = 3 if N > 64 else 8
    rng = np.random.default_rng(600 + i)
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    y0 = rng.integers(0, 3, size=(B, N)).astype(np.float64) if offlane else np.zeros((B, N))
    v0 = np.full((B, N), 1.7) if cfg.mobility_vary else rng.uniform(1.1, 2.7, size=(B, N))
    dt = torch.float64 if f64 else torch.float32
    fused, two = make_env(cfg, B, dtype=dt), make_env(cfg, B, dtype=dt)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    for e in (fused, two):
        e.reset_topology(x0, y0, v0)
    orc.reset(x0, y0, v0)
    acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
    exp_rew = cfg.reward_design in (3, 4) and mode != STEP_DESIGN
    fam = KERNEL_FAST64 if N <= 64 else KERNEL_WIDE
    call = {STEP_MY_STEP: two.my_step, STEP_MY_STEP_CH: two.my_step_ch, STEP_DESIGN: two.my_step_design}[mode]

    def same(got, want, exp_tol):
        want = want if f64 else want.astype(np.float32)
        if exp_tol:
            return bool(np.all(np.abs(got.astype(np.float64) - want) <= (EXP_ATOL if f64 else 1e-6)))
        return np.array_equal(got, want)

    for t in range(24):
        new = rng.integers(0, A, size=(B, N))
        acts = np.where(rng.random((B, N)) < sticky, acts, new).astype(np.int32)
        ep, eps = float(t // 5), 0.99 ** t
        a_dev = fused._actions(acts)
        obs, rew, _ = fused._step(mode, a_dev, t, ep, eps, want_chobs=True)
        assert (fused.last_kernel() & 15) == fam, fused.last_kernel()
        chobs2, rew2 = call(a_dev, t)
        k2 = two.last_kernel()
        st2 = two.obtain_state(chobs2, a_dev, rew2, ep, eps)
        assert two.last_kernel() == k2 and (k2 & 15) == fam             # no second launch
        o_rew, o_chobs = orc.step(mode, acts, t)
        o_state = orc.obtain_state(acts, o_chobs, o_rew, ep, eps)
        torch.cuda.synchronize()
        assert same(rew.cpu().numpy(), o_rew, exp_rew) and same(rew2.cpu().numpy(), o_rew, exp_rew), t
        assert same(fused._chobs.cpu().numpy(), o_chobs, False) and same(chobs2.cpu().numpy(), o_chobs, False), t
        tol = exp_rew and cfg.State.add_reward
        assert same(obs.cpu().numpy(), o_state, tol), (t, np.argwhere(obs.cpu().numpy() != (o_state if f64 else o_state.astype(np.float32)))[:5])
        assert same(st2.cpu().numpy(), o_state, tol), t
        if t % 9 == 8:
            d = rng.integers(1, 4, size=(B, N)).astype(np.uint8)
            for e in (fused, two, orc):
                e.update_velocity(d)
    st, oe = fused.export_state(), orc.export()
    assert np.array_equal(st["seq"].cpu().numpy(), oe["seq"])
    assert np.array_equal(st["x"].cpu().numpy(), oe["x"])
    assert np.array_equal(st["pos_x"].cpu().numpy(), oe["pos_x"])
    if cfg.track_arrival:
        assert np.array_equal(st["la"].cpu().numpy().astype(np.int64), oe["la"])
    fused.check()
    two.check()
