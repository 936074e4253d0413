"""GPU parity of the three-launch form (csrc/step_large.hpp): the path that serves what the reference takes and the
one-workgroup kernels do not - num_users > 256, num_channels > 256, State.num_bins > 64 (TestEnv has no limits,
test_env.py:12-13, 40).  Two angles:
  * every reference fixture replayed through it (DIRAL_PATH_LARGE on small handles): the same bar as the specialised
    kernels - bit-exact against the oracle, 1 ulp (pow) against the reference's own numbers;
  * sizes only it runs (300 ... 2048 vehicles, up to 700 resources, up to 300 bins) against the oracle, every step
    kind, the State flags, PRR / arrival tracking, velocity updates, lanes off the y = 0 line, the secondary
    observation modes.
"""
import numpy as np
import pytest
import torch

import tests.test_gpu_parity as tp
from diral_amd.config import (KERNEL_LARGE, STEP_DESIGN, STEP_MY_STEP, STEP_MY_STEP_CH, bench_config, c2_config)
from tests.golden_util import golden_names

pytestmark = pytest.mark.gpu


@pytest.fixture
def large_path(monkeypatch):
    """make_env of the parity tests hands out handles pinned to the large path."""
    plain = tp.make_env

    def make(cfg, B, mode=STEP_MY_STEP, dtype=torch.float64):
        env = plain(cfg, B, mode, dtype)
        env.force_large_path()
        return env
    monkeypatch.setattr(tp, "make_env", make)


@pytest.mark.parametrize("name", golden_names())
def test_golden_replay_on_the_large_path(name, large_path):
    tp.test_golden_replay_on_gpu(name)


@pytest.mark.parametrize("mode,rd", [(STEP_MY_STEP, 1), (STEP_MY_STEP, 2), (STEP_MY_STEP, 3), (STEP_MY_STEP, 5),
                                     (STEP_MY_STEP_CH, 2), (STEP_MY_STEP_CH, 4), (STEP_DESIGN, 2)])
def test_c2_shapes_on_the_large_path(mode, rd):
    """64 vehicles / 32 resources, rich State flags, sticky actions, PRR tracking: against the oracle."""
    cfg = c2_config(reward_design=rd, State=dict(add_channel_obs=True, add_reward=True, add_index=True, add_velocity=True,
                                                  add_position=True))
    tp.random_rollout(cfg, B=12, T=30, seed=900 + rd + 10 * mode, mode=mode, sticky=0.7, track_prr=(mode == STEP_MY_STEP),
                      path="large", expect_kernel=KERNEL_LARGE)


def test_c5_shapes_with_velocity_updates_on_the_large_path():
    tp.random_rollout(bench_config(128, 64, 4000.0, mobility_vary=True, proportional_fair=True), B=6, T=55, seed=907,
                      vel_every=25, sticky=0.9, path="large", expect_kernel=KERNEL_LARGE)


def _lanes(rng, B, N):
    return rng.integers(0, 2, size=(B, N)).astype(np.float64) * 1.5


@pytest.mark.parametrize("N,A,K,L,rc,mode,kw", [
    (257, 8, 20, 4000.0, 250.0, STEP_MY_STEP, {}),                               # one vehicle past the limit
    (300, 40, 20, 5000.0, 250.0, STEP_MY_STEP_CH, dict(reward_design=3)),
    (300, 40, 20, 5000.0, 120.0, STEP_DESIGN, {}),
    (512, 64, 20, 8000.0, 250.0, STEP_MY_STEP, dict(reward_design=1)),
    (700, 300, 20, 9000.0, 250.0, STEP_MY_STEP, dict(reward_design=5)),          # more resources than any other kernel takes
    (1024, 64, 40, 16000.0, 250.0, STEP_MY_STEP, {}),
    (200, 200, 64, 3000.0, 250.0, STEP_MY_STEP, {}),                             # below the sizes, beyond the general kernel's LDS
    (100, 16, 300, 2000.0, 250.0, STEP_MY_STEP, {}),                             # bins only
    (64, 700, 20, 2000.0, 250.0, STEP_MY_STEP_CH, dict(reward_design=2)),        # resources only
])
def test_sizes_only_the_large_path_runs(N, A, K, L, rc, mode, kw):
    cfg = bench_config(N, A, L, communication_range=rc,
                       State=dict(num_bins=K, add_channel_obs=True, add_reward=True, add_index=True, add_position=True), **kw)
    T = 8 if N >= 700 else 14
    tp.random_rollout(cfg, B=2 if N >= 700 else 3, T=T, seed=1000 + N + A, mode=mode, sticky=0.5,
                      track_prr=(mode == STEP_MY_STEP), expect_kernel=KERNEL_LARGE)


def test_lanes_off_the_line_and_velocity_updates_at_400_vehicles():
    cfg = bench_config(400, 50, 6000.0, mobility_vary=True, State=dict(add_velocity=True, add_position=True))
    tp.random_rollout(cfg, B=2, T=30, seed=1400, vel_every=25, y0=_lanes, expect_kernel=KERNEL_LARGE)


def test_secondary_observation_modes_at_300_vehicles():
    """add_positional_dist (sorted true distances) and the type-1 histogram beyond 256 vehicles: posdist_kernel.hpp's
    flat ranking and its literal statement."""
    cfg = bench_config(300, 16, 5000.0, State=dict(add_positional_dist=True, add_positional_dist_type=1, num_bins=12))
    tp.random_rollout(cfg, B=2, T=8, seed=1500, expect_kernel=KERNEL_LARGE)
    cfg = bench_config(300, 16, 5000.0, State=dict(add_positional_dist=True))
    tp.random_rollout(cfg, B=2, T=8, seed=1501, y0=_lanes, expect_kernel=KERNEL_LARGE)


def test_2048_vehicles_two_slots():
    """The table column of 2048 viewers in one wave's LDS (two columns per workgroup)."""
    cfg = bench_config(2048, 128, 30000.0)
    tp.random_rollout(cfg, B=1, T=3, seed=1600, expect_kernel=KERNEL_LARGE)


def test_limits_of_the_large_path():
    from diral_amd.config import ERR_UNSUPPORTED
    from diral_amd.vec_env import DiralError, VecV2VEnv
    for cfg in (bench_config(4097, 8, 50000.0), bench_config(64, 4097, 2000.0),
                bench_config(64, 300, 2000.0, State=dict(piggybacking=True, add_channel_obs=True)),
                bench_config(2000, 8, 30000.0, State=dict(add_positional_dist_type=1))):
        with pytest.raises(Exception) as ei:
            VecV2VEnv(cfg, batch=1, device="cuda:0")
        assert getattr(ei.value, "status", ERR_UNSUPPORTED) == ERR_UNSUPPORTED, ei.value
    env = VecV2VEnv(bench_config(4096, 16, 60000.0), batch=1, device="cuda:0")     # the largest env: 16.7 M table entries
    rng = np.random.default_rng(5)
    x0 = rng.integers(0, 60000, size=(1, 4096)).astype(np.float64)
    env.reset_topology(x0, 0.0, rng.uniform(1.1, 2.7, size=(1, 4096)))
    obs, rew, done = env.step(rng.integers(0, 16, size=(1, 4096)).astype(np.int32), 0)
    torch.cuda.synchronize()
    env.check()
    assert env.last_kernel() == KERNEL_LARGE
    assert obs.shape == (1, 4096, 36) and float(obs[0, :, :16].sum()) == 4096.0


@pytest.mark.parametrize("N,A,L", [(40, 6, 1500.0), (300, 20, 6000.0), (1100, 30, 20000.0)])
def test_entries_a_million_stamps_old_take_the_64_bit_merge(N, A, L):
    """large_merge2_kernel orders entries by a 20-bit rank (how far the entry lags its subject); imported tables with
    entries 2^20 stamps behind make the column pair fall back on the (number, source) keys of large_merge_column - and
    beyond 1024 vehicles that form is the only one.  Tables, state and rewards against the oracle over 6 slots."""
    from oracle.oracle import Oracle, SQ_IEEE
    cfg = bench_config(N, A, L)
    B = 2
    rng = np.random.default_rng(4242 + N)
    pos_x = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    vel = rng.uniform(1.1, 2.7, size=(B, N))
    T0 = 1_300_000
    kind = rng.integers(0, 4, size=(B, N, N))                       # 0 never heard, 1 fresh, 2 a few stamps old, 3 ancient
    lag = np.where(kind == 1, rng.integers(1, 4, size=(B, N, N)), np.where(kind == 2, rng.integers(4, 40, size=(B, N, N)),
                                                                            rng.integers(1 << 20, 1_200_000, size=(B, N, N))))
    seq = np.where(kind == 0, 0, T0 - lag).astype(np.int32)
    # equal (subject, number) => equal xpos (what a run produces and import_state asks for): xpos a function of both
    kk = np.broadcast_to(np.arange(N)[None, None, :], (B, N, N))
    x = np.where(seq > 0, (kk * 7919 + seq.astype(np.int64) * 31) % int(L), 0).astype(np.float64)
    age = np.where(kind == 0, 0, np.minimum(lag, 255)).astype(np.int32)
    for u in range(N):
        seq[:, u, u] = T0
        age[:, u, u] = 0
        x[:, u, u] = pos_x[:, u]
    env = tp.make_env(cfg, B)
    env.force_large_path()
    env.reset_topology(pos_x, 0.0, vel)
    env.import_state(pos_x, np.zeros((B, N)), vel, seq=seq, age=age, x=x)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=4)
    orc.reset(pos_x, np.zeros((B, N)), vel)
    orc.import_state(seq=seq, age=age, x=x, y=np.zeros((B, N, N)))
    for t in range(6):
        acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, chobs, _ = tp.gpu_step(env, STEP_MY_STEP, acts, t)
        assert env.last_kernel() == KERNEL_LARGE
        o_rew, o_chobs = orc.step(STEP_MY_STEP, acts, t)
        assert np.array_equal(rew, o_rew) and np.array_equal(chobs, o_chobs)
        assert np.array_equal(obs, orc.obtain_state(acts, o_chobs, o_rew)), t
        st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        oe = orc.export()
        assert np.array_equal(st["seq"], oe["seq"]) and np.array_equal(st["x"], oe["x"]), t
        assert np.array_equal(st["age"], np.minimum(oe["age"], 255)), t
    env.check()


def test_float32_outputs_and_a_closed_loop_on_a_320_vehicle_handle():
    """float32 outputs = the float32 cast of the float64 ones (the arithmetic is float64 either way); the SPS closed
    loop (`step_policy`, here the three-launch form: env step with the channel observation, reward shaping, the agents'
    decisions) and a rollout captured into a hipGraph run on such a handle like on any other."""
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    N, A, B = 320, 24, 3
    cfg = bench_config(N, A, 6000.0, State=dict(add_channel_obs=True, add_reward=True, add_position=True))
    e64 = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float64)
    e32 = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32)
    for e in (e64, e32):
        e.reset_topology(seed=3)
    for t in range(12):
        a = e64.sample(50 + t)
        o64, r64, _ = e64.step(a, t)
        o32, r32, _ = e32.step(a, t)
        assert e32.last_kernel() == KERNEL_LARGE
        assert torch.equal(o32, o64.to(torch.float32)) and torch.equal(r32, r64.to(torch.float32)), t
    runs = []
    for capture in (False, True):
        env = VecV2VEnv(bench_config(N, A, 6000.0), batch=B, device="cuda:0", io_ring=2)
        env.reset_topology(seed=8)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=6)
        runs.append((env, pol, GraphRollout(env, pol, K=4, capture=capture, fused=True)))
    (e1, p1, r1), (e2, p2, r2) = runs
    r1.run(5)
    r2.run(4)                                                        # (the capture itself ran the K slots once)
    torch.cuda.synchronize()
    assert r1.slots == r2.slots == 20
    assert e1.last_kernel() == KERNEL_LARGE
    a, b = e1.export_state(), e2.export_state()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.allclose(e1.metrics(), e2.metrics(), rtol=1e-12, atol=1e-9)
    for e, _, r in runs:
        e.check()
        r.close()


@pytest.mark.parametrize("i", range(48))
def test_random_configuration_on_the_large_path(i):
    """tests/test_gpu_fuzz.py's generator - every State flag, reward design, step kind, static and design topologies,
    State blocks without piggybacked tables, both secondary observation modes, proportional fairness - with the handle
    pinned to the three-launch form: against the oracle, PRR metrics tracked on odd draws."""
    from diral_amd.config import ConfigError
    from tests.test_gpu_fuzz import draw_case
    cfg, mode, sticky, vel_every = draw_case(i)
    if not (cfg.mobility or cfg.enable_design_topology):
        cfg = cfg.replace(enable_design_topology=True)
    cfg.validate()
    tp.random_rollout(cfg, B=3 if cfg.num_users > 64 else 8, T=20, seed=300 + i, mode=mode, sticky=sticky,
                      vel_every=vel_every or None, threads=8, track_prr=bool(i & 1), path="large", expect_kernel=KERNEL_LARGE)


def draw_large_case(i):
    rng = np.random.default_rng(12000 + i)
    N = int(rng.choice([257, 300, 333, 400, 512, 513, 640]))
    A = int(rng.choice([1, 5, 32, 64, 100, 257, 300]))
    K = int(rng.choice([1, 20, 64, 65, 130]))
    mode = int(rng.choice([STEP_MY_STEP, STEP_MY_STEP, STEP_MY_STEP_CH, STEP_DESIGN]))
    rd = int(rng.choice([2, 3, 4])) if mode == STEP_MY_STEP_CH else int(rng.integers(1, 6))
    L = float(rng.choice([8.0, 20.0, 40.0]) * N + rng.integers(20, 200))
    state = dict(type=2, add_reward=bool(rng.random() < 0.4), add_action=bool(rng.random() < 0.8),
                 add_index=bool(rng.random() < 0.3), add_velocity=bool(rng.random() < 0.3),
                 action_index=str(rng.choice(["binary", "real"])), add_position=bool(rng.random() < 0.3),
                 add_positional_dist=bool(rng.random() < 0.2), add_positional_dist_piggy=bool(rng.random() < 0.85),
                 add_positional_dist_type=int(rng.choice([1, 2, 2, 2])), num_bins=K,
                 add_channel_obs=bool(rng.random() < 0.4))
    cfg = bench_config(N, A, L, reward_design=rd, State=state, mobility_vary=bool(rng.random() < 0.4),
                       enable_fingerprint=bool(rng.random() < 0.3),
                       proportional_fair=bool(rng.random() < 0.2 and mode == STEP_MY_STEP),
                       communication_range=float(rng.choice([30.0, 120.0, 250.0])),
                       bin_range=float(rng.choice([100.0, 500.0])))
    return cfg, mode, float(rng.choice([0.0, 0.5, 0.9])), int(rng.choice([0, 5]))


@pytest.mark.parametrize("i", range(24))
def test_random_configuration_beyond_256_vehicles(i):
    cfg, mode, sticky, vel_every = draw_large_case(i)
    cfg.validate()
    tp.random_rollout(cfg, B=2, T=11, seed=700 + i, mode=mode, sticky=sticky, vel_every=vel_every or None, threads=8,
                      track_prr=bool(i & 1), expect_kernel=KERNEL_LARGE)


@pytest.mark.parametrize("N,A,L", [(300, 5, 240.0), (520, 3, 200.0)])
def test_piggybacking_beyond_256_vehicles(N, A, L):
    """State.piggybacking (test_env.py:241-254) is defined while every receiver hears a transmitter on every used resource -
    a highway no longer than the communication range; the A * A observation of piggyback_kernel.hpp around the
    three-launch step, against the oracle (which replays np.insert on real arrays)."""
    cfg = bench_config(N, A, L, communication_range=L + 1.0, State=dict(piggybacking=True, add_channel_obs=True, add_reward=True))
    assert cfg.chobs_width == A * A
    tp.random_rollout(cfg, B=2, T=6, seed=2000 + N, sticky=0.3, expect_kernel=KERNEL_LARGE)
