"""GPU parity: the HIP path (through the C-ABI) vs the reference goldens and
vs the CPU oracle, on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact for collision / indexing work
(collision counts, rewards of the integer-valued designs, one-hot, histogram
bins, sequence numbers, ages, arrival stamps, information-age histogram) and for
positions; distances within 1 ulp of the reference (the reference's `**2` is
libm pow(), the kernel uses the correctly rounded x*x - DESIGN.md), exp()-based
rewards within 2e-15 absolute (1-2 ulp of exp itself).  Against the oracle in its IEEE-square mode everything
except exp() is bit-exact.
"""
import numpy as np
import pytest
import torch

from diral_amd.config import (EnvConfig, STEP_DESIGN, STEP_MY_STEP, STEP_MY_STEP_CH, bench_config,
                              c2_config)
from tests.golden_util import Golden, golden_names, ulp_diff

pytestmark = pytest.mark.gpu

EXP_ATOL = 2e-15   # device exp() vs glibc exp(): <= 1-2 ulp of exp(x) <= e; the rewards built
                   # from it (1-exp(1-R), test_env.py:414) cancel, so the bound is absolute
DIST_ULP = 1       # sqrt(x*x+y*y) vs sqrt(pow(x,2)+pow(y,2))

UNBUILT = set()


def make_env(cfg, B, mode=STEP_MY_STEP, dtype=torch.float64):
    from diral_amd.vec_env import VecV2VEnv
    return VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=dtype, step_mode=mode)


def gpu_step(env, mode, acts, t, ep=0.0, eps=1.0):
    a = env._actions(np.asarray(acts))
    obs, rew, done = env._step(mode, a, t, ep, eps, want_chobs=True)
    torch.cuda.synchronize()
    return obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), env._chobs.cpu().numpy().copy(), \
        done.cpu().numpy().copy()


def uses_exp(cfg, mode):
    return cfg.reward_design in (3, 4)


def exp_close(a, b):
    return bool(np.all(np.abs(np.asarray(a) - np.asarray(b)) <= EXP_ATOL))


@pytest.mark.parametrize("name", [n for n in golden_names() if n not in UNBUILT])
def test_golden_replay_on_gpu(name):
    """Every reference fixture replayed through libdiral_env.so (B=3 replicas)."""
    from oracle.oracle import Oracle, SQ_IEEE
    g = Golden(name)
    B = 3
    env = make_env(g.cfg, B)
    env.reset_topology(g["x0"], g["y0"], g["v0"])
    orc = Oracle(g.cfg, batch=1, sq_mode=SQ_IEEE)
    orc.reset(g["x0"], g["y0"], g["v0"])
    ck = g.table_checkpoints()
    S = env.S
    cfg = g.cfg
    for i, mode, acts, t, (ep, eps) in g.steps():
        obs, rew, chobs, done = gpu_step(env, mode, acts, t, ep, eps)
        o_rew, o_chobs = orc.step(mode, acts, t)
        o_state = orc.obtain_state(acts, o_chobs, o_rew, ep, eps)
        for b in range(B):
            # --- vs the oracle (IEEE squares): bit-exact except exp() rewards
            if uses_exp(cfg, mode):
                assert exp_close(rew[b], o_rew[0]), (name, i)
            else:
                assert np.array_equal(rew[b], o_rew[0]), (name, i, rew[b], o_rew[0])
            assert np.array_equal(chobs[b], o_chobs[0]), (name, i)
            if cfg.State.add_reward and uses_exp(cfg, mode):
                assert exp_close(obs[b], o_state[0])
            else:
                assert np.array_equal(obs[b], o_state[0]), (name, i, np.argwhere(obs[b] != o_state[0])[:5])
            # --- vs the reference fixture
            ref_rew, ref_chobs, ref_state = g["rews"][i], g["chobs"][i], g["state"][i]
            assert exp_close(rew[b], ref_rew) if uses_exp(cfg, mode) else np.array_equal(rew[b], ref_rew), (name, i)
            assert ulp_diff(chobs[b], ref_chobs) <= DIST_ULP, (name, i)
            assert ulp_diff(obs[b], ref_state) <= DIST_ULP or (cfg.State.add_reward and exp_close(obs[b], ref_state)), (name, i)
            if cfg.State.add_positional_dist_piggy:
                K = cfg.State.num_bins
                off = (cfg.num_channels if cfg.State.action_index == "binary" else 1) if cfg.State.add_action else 0
                off += cfg.chobs_width if cfg.State.add_channel_obs else 0
                off += cfg.num_users - 1 if cfg.State.add_positional_dist else 0
                if cfg.State.add_positional_dist_type == 2:
                    assert np.array_equal(obs[b][:, off:off + K], ref_state[:, off:off + K]), "histogram bins"
            assert done[b] == ((t % cfg.episode_interval) == cfg.episode_interval - 1)
        if i in g.vel_updates:
            env.update_velocity(g.vel_updates[i])
            orc.update_velocity(g.vel_updates[i])
        if g.trace is not None and i == g.trace_after:
            env.load_saved_positions(g.trace)
            orc.set_trace(g.trace)
        ia = env.info_age(t).cpu().numpy()
        st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        oe = orc.export()
        for b in range(B):
            assert np.array_equal(st["pos_x"][b], g["pos_x"][i]), (name, i)
            assert np.array_equal(st["vel"][b], g["vel"][i]), (name, i)
            assert np.array_equal(ia[b], g["ia"][i]), (name, i)
            assert np.array_equal(st["seq"][b], oe["seq"][0]), (name, i)
            assert np.array_equal(st["age"][b], np.minimum(oe["age"][0], 255)), (name, i)
            assert np.array_equal(st["x"][b], oe["x"][0]), (name, i)
            assert np.array_equal(st["y"][b], oe["y"][0]), (name, i)
            assert np.array_equal(st["la"][b].astype(np.int64), oe["la"][0]), (name, i)
            if i in ck:
                j = ck[i]
                assert np.array_equal(st["seq"][b], g["tab_seq"][j])
                assert np.array_equal(st["age"][b], np.minimum(g["tab_age"][j], 255))
                assert np.array_equal(st["x"][b], g["tab_x"][j])
                assert np.array_equal(st["y"][b], g["tab_y"][j])
                assert np.array_equal(st["la"][b], g["tab_la"][j])
    env.check()
    if cfg.State.piggybacking:
        # TestEnv.prev_obs after the last slot (test_env.py:260-261): distances, within 1 ulp of the reference's pow()
        po = env.prev_obs().cpu().numpy()
        for b in range(B):
            assert np.array_equal(po[b], orc.prev_obs()[0]) and ulp_diff(po[b], g["prev_obs"]) <= DIST_ULP
        if "keyerror_actions" in g.d.files:
            # the slot the reference left with KeyError (`self.prev_obs[None]`, test_env.py:243): the sticky device flag
            from diral_amd.config import ERR_PIGGY_NO_TX
            from diral_amd.vec_env import DiralError
            gpu_step(env, STEP_MY_STEP, g["keyerror_actions"], int(g["keyerror_t"]))
            with pytest.raises(DiralError) as ei:
                env.check()
            assert ei.value.status == ERR_PIGGY_NO_TX
            env.check()                                             # reported once, then cleared


def random_rollout(cfg, B, T, seed, mode=STEP_MY_STEP, sticky=0.0, vel_every=None, threads=8, track_prr=False,
                   expect_kernel=None, force_general=False, path=None, y0=None):
    """GPU vs oracle (IEEE squares) on B different random envs, T slots; returns
    the number of compared slots.  Everything must match bit for bit (exp()
    rewards within EXP_ATOL).  The specialised kernels serve every configuration they can
    (`expect_kernel`: assert which family ran); `force_general` pins the run to the general
    kernel (DIRAL_OPT_KERNEL_PATH), `path="large"` to the three launches of csrc/step_large.hpp (the only path beyond
    256 vehicles / 256 resources / 64 bins); track_prr: PRR metrics also in my_step (a build extension); `y0`: a
    function (rng, B, N) -> lanes (default: the one-lane highway the reference draws)."""
    from oracle.oracle import Oracle, SQ_IEEE
    rng = np.random.default_rng(seed)
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    cfg = cfg.replace(track_arrival=True, track_prr=track_prr)
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    y0 = np.zeros((B, N)) if y0 is None else y0(rng, B, N)
    v0 = np.full((B, N), 1.7) if cfg.mobility_vary else rng.uniform(1.1, 2.7, size=(B, N))
    env = make_env(cfg, B)
    env.force_general_kernel(force_general)
    if path == "large":
        env.force_large_path()
    env.reset_topology(x0, y0, v0)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=threads)
    orc.reset(x0, y0, v0)
    acts = rng.integers(0, A, size=(B, N))
    for t in range(T):
        new = rng.integers(0, A, size=(B, N))
        acts = np.where(rng.random((B, N)) < sticky, acts, new).astype(np.int32)
        obs, rew, chobs, _ = gpu_step(env, mode, acts, t)
        if expect_kernel is not None:
            assert (env.last_kernel() & 15) == expect_kernel, env.last_kernel()
        o_rew, o_chobs = orc.step(mode, acts, t)
        o_state = orc.obtain_state(acts, o_chobs, o_rew)
        if uses_exp(cfg, mode):
            assert exp_close(rew, o_rew)
        else:
            assert np.array_equal(rew, o_rew), t
        assert np.array_equal(chobs, o_chobs), t
        if uses_exp(cfg, mode) and cfg.State.add_reward:       # the state carries the exp() reward
            assert exp_close(obs, o_state), t
        else:
            assert np.array_equal(obs, o_state), (t, np.argwhere(obs != o_state)[:5])
        if t % 6 == 4 and env.S > 0:
            # a stand-alone obtain_state with FOREIGN arguments (test_env.py:527-583 on the current tables):
            # diral_env_observe -> observe_kernel.hpp (the general kernel's observe mode when forced)
            from diral_amd.config import KERNEL_GENERAL, KERNEL_LARGE, KERNEL_OBSERVE
            fa = rng.integers(0, A, size=(B, N)).astype(np.int32)
            fc = rng.uniform(0.0, 300.0, size=(B, N, cfg.chobs_width))
            fr = rng.uniform(-3.0, 1.0, size=(B, N))
            s1 = env.obtain_state(fc, fa, fr, 3.0, 0.25).cpu().numpy()
            assert (env.last_kernel() & 15) == (KERNEL_LARGE if (path == "large" or expect_kernel == KERNEL_LARGE) else (KERNEL_GENERAL if force_general else KERNEL_OBSERVE))
            s2 = orc.obtain_state(fa, fc, fr, 3.0, 0.25)
            assert np.array_equal(s1, s2), (t, np.argwhere(s1 != s2)[:5])
        if vel_every and t % vel_every == vel_every - 1:
            draws = rng.integers(1, 4, size=(B, N)).astype(np.uint8)
            env.update_velocity(draws)
            orc.update_velocity(draws)
    st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
    oe = orc.export()
    assert np.array_equal(st["pos_x"], oe["pos_x"])
    assert np.array_equal(st["vel"], oe["vel"])
    assert np.array_equal(st["seq"], oe["seq"])
    assert np.array_equal(st["age"], np.minimum(oe["age"], 255))
    assert np.array_equal(st["x"], oe["x"])
    assert np.array_equal(st["y"], oe["y"])
    assert np.array_equal(st["la"].astype(np.int64), oe["la"])
    assert np.array_equal(env.info_age(T - 1).cpu().numpy(), orc.info_age(T - 1))
    if cfg.proportional_fair and mode == STEP_MY_STEP and sticky >= 0.9 and T >= 40:
        assert oe["pf"].max() > 10                    # the threshold was crossed: the penalty branch ran
    m, om = env.metrics().cpu().numpy(), orc.metrics()
    assert np.array_equal(m[:, [0, 2, 3]], om[:, [0, 2, 3]])                # counts: exact
    assert np.allclose(m[:, 1], om[:, 1], rtol=1e-12, atol=1e-9)            # float sums: order differs
    if track_prr or mode == STEP_MY_STEP_CH:
        assert np.array_equal(m[:, 5], om[:, 5])
        assert np.allclose(m[:, 4], om[:, 4], rtol=1e-12, atol=1e-9)
        # PRR parity (north_star: within 1e-6)
        prr = m[:, 4] / np.maximum(m[:, 5], 1)
        oprr = om[:, 4] / np.maximum(om[:, 5], 1)
        assert np.max(np.abs(prr - oprr)) < 1e-6
    env.check()
    return T


def _fam(N):
    from diral_amd.config import KERNEL_FAST64, KERNEL_WIDE
    return KERNEL_FAST64 if N <= 64 else KERNEL_WIDE


def test_c2_random_vs_oracle():
    """The headline configuration on the kernel the bench runs (step_fast64, RICH: the rollout asks for
    the channel observation too), PRR metrics tracked in my_step as well."""
    random_rollout(c2_config(), B=96, T=70, seed=1, expect_kernel=_fam(64))
    random_rollout(c2_config(), B=32, T=40, seed=21, track_prr=True, expect_kernel=_fam(64))


def test_c2_random_vs_oracle_general_kernel():
    from diral_amd.config import KERNEL_GENERAL
    random_rollout(c2_config(), B=48, T=50, seed=1, track_prr=True, force_general=True, expect_kernel=KERNEL_GENERAL)


def test_c2_ch_mode_vs_oracle():
    random_rollout(c2_config(reward_design=3), B=48, T=40, seed=2, mode=STEP_MY_STEP_CH, sticky=0.6)


def test_c2_design_mode_vs_oracle():
    random_rollout(c2_config(), B=32, T=30, seed=3, mode=STEP_DESIGN)


def test_c2_channel_obs_and_flags_vs_oracle():
    cfg = c2_config(reward_design=1, enable_fingerprint=False,
                    State=dict(add_channel_obs=True, add_reward=True, add_index=True, add_velocity=True,
                               add_position=True))
    random_rollout(cfg, B=24, T=30, seed=4, sticky=0.8)


def test_c3_congested_vs_oracle():
    random_rollout(bench_config(256, 64, 4000.0), B=6, T=26, seed=5, expect_kernel=_fam(256))
    random_rollout(bench_config(256, 64, 4000.0), B=4, T=26, seed=25, track_prr=True, expect_kernel=_fam(256))


def test_c5_dynamic_density_vs_oracle():
    random_rollout(bench_config(128, 64, 4000.0, mobility_vary=True), B=10, T=55, seed=6, vel_every=25,
                   expect_kernel=_fam(128))
    random_rollout(bench_config(128, 64, 4000.0, mobility_vary=True), B=6, T=30, seed=26, vel_every=25, track_prr=True,
                   expect_kernel=_fam(128))


def test_c3_c5_vs_oracle_general_kernel():
    from diral_amd.config import KERNEL_GENERAL
    random_rollout(bench_config(256, 64, 4000.0), B=4, T=20, seed=5, track_prr=True, force_general=True,
                   expect_kernel=KERNEL_GENERAL)
    random_rollout(bench_config(128, 64, 4000.0, mobility_vary=True), B=6, T=30, seed=6, vel_every=25, track_prr=True,
                   force_general=True, expect_kernel=KERNEL_GENERAL)


@pytest.mark.parametrize("N,A,L", [(1, 1, 50.0), (2, 5, 100.0), (63, 7, 900.0), (65, 3, 900.0),
                                   (129, 40, 3000.0), (200, 13, 3000.0), (256, 128, 5000.0)])
def test_ragged_sizes_vs_oracle(N, A, L):
    random_rollout(bench_config(N, A, L, communication_range=180.0), B=4, T=24, seed=100 + N)


def test_rd5_and_proportional_fair_vs_oracle():
    random_rollout(c2_config(reward_design=5, proportional_fair=True), B=16, T=40, seed=7, sticky=0.95)


@pytest.mark.parametrize("N,A", [(64, 32), (20, 4), (128, 64), (256, 16), (90, 9)])
@pytest.mark.parametrize("mode", [STEP_MY_STEP, STEP_MY_STEP_CH, STEP_DESIGN])
def test_proportional_fairness_on_the_specialised_kernels(N, A, mode):
    """test_env.py:215-222: a transmitter that collided more than pf_threshold (10) slots in a row
    is paid pf_penalty (-10), a success resets its counter - only in my_step.  Sticky actions so the
    counters pass the threshold; specialised kernels (RICH) vs the oracle, incl. the pf counters."""
    from diral_amd.config import KERNEL_FAST64, KERNEL_WIDE
    L = 12.0 * N + 100
    cfg = bench_config(N, A, L, reward_design=2 if mode != STEP_MY_STEP else 1, proportional_fair=True)
    random_rollout(cfg, B=6, T=45, seed=900 + N + mode, mode=mode, sticky=0.97, track_prr=False,
                   expect_kernel=KERNEL_FAST64 if N <= 64 else KERNEL_WIDE)


def test_f32_output_is_cast_of_f64():
    cfg = c2_config(State=dict(add_channel_obs=True))
    rng = np.random.default_rng(9)
    B = 8
    x0 = rng.integers(0, 2000, size=(B, 64)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, 64))
    e64, e32 = make_env(cfg, B, dtype=torch.float64), make_env(cfg, B, dtype=torch.float32)
    e64.reset_topology(x0, 0.0, v0)
    e32.reset_topology(x0, 0.0, v0)
    for t in range(25):
        a = rng.integers(0, 32, size=(B, 64)).astype(np.int32)
        o64, r64, _ = e64.step(a, t)
        o32, r32, _ = e32.step(a, t)
        torch.cuda.synchronize()
        assert torch.equal(o64.to(torch.float32), o32)
        assert torch.equal(r64.to(torch.float32), r32)


def test_observe_equals_fused_step_observation():
    cfg = c2_config(State=dict(add_channel_obs=True, add_reward=True))
    rng = np.random.default_rng(10)
    B = 5
    env = make_env(cfg, B)
    env.reset_topology(rng.integers(0, 2000, size=(B, 64)).astype(np.float64), 0.0,
                       rng.uniform(1.1, 2.7, size=(B, 64)))
    for t in range(30):
        a = rng.integers(0, 32, size=(B, 64)).astype(np.int32)
        obs, rew, chobs, _ = gpu_step(env, STEP_MY_STEP, a, t)
        again = env.obtain_state(chobs, a, rew).cpu().numpy()
        assert np.array_equal(obs, again)


def test_reference_named_two_call_sequence():
    """`obs, rews = env.my_step(a, t); state = env.obtain_state(obs, a, rews)`
    (main_test.py:144-164) gives the same state as the fused step()."""
    cfg = c2_config()
    rng = np.random.default_rng(11)
    B = 4
    x0 = rng.integers(0, 2000, size=(B, 64)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, 64))
    fused, split = make_env(cfg, B), make_env(cfg, B)
    fused.reset_topology(x0, 0.0, v0)
    split.reset_topology(x0, 0.0, v0)
    for t in range(22):
        a = rng.integers(0, 32, size=(B, 64)).astype(np.int32)
        o1, r1, _ = fused.step(a, t)
        chobs, rews = split.my_step(a, t)
        o2 = split.obtain_state(chobs, a, rews)
        torch.cuda.synchronize()
        assert torch.equal(o1, o2)
        assert torch.equal(r1, rews)


def test_bad_action_is_flagged():
    from diral_amd.vec_env import DiralError
    env = make_env(c2_config(), 2)
    env.reset_topology(seed=3)
    a = np.zeros((2, 64), np.int32)
    a[1, 5] = 32
    env.step(a, 0)
    with pytest.raises(DiralError) as ei:
        env.check()
    assert ei.value.status == -6
    env.step(np.zeros((2, 64), np.int32), 1)
    env.check()


def test_device_topology_draws_follow_reference_distribution():
    """network.py:103-110: x integer in [0,L), y = 0, v in [1.1,2.7) (1.7 if vary)."""
    env = make_env(c2_config(), 512)
    env.reset_topology(seed=42)
    st = env.export_state(tables=False)
    x, y, v = st["pos_x"], st["pos_y"], st["vel"]
    assert torch.all(x == torch.floor(x)) and x.min() >= 0 and x.max() < 2000
    assert torch.all(y == 0)
    assert v.min() >= 1.1 and v.max() < 2.7 and abs(v.mean().item() - 1.9) < 0.02
    assert abs(x.mean().item() - 999.5) < 15
    a = env.sample(seed=1)
    assert a.min() >= 0 and a.max() < 32
    cnt = torch.bincount(a.flatten().long(), minlength=32).double()
    assert (cnt / cnt.sum() - 1 / 32).abs().max() < 0.01
    env2 = make_env(c2_config(mobility_vary=True), 4)
    env2.reset_topology(seed=1)
    assert torch.all(env2.export_state(tables=False)["vel"] == 1.7)


def test_full_size_c2_properties_and_sampled_oracle():
    """BASELINE configs[1] at full size: B=4096 x 64 UE x 32 res.  Size-
    independent properties on every env + bit-exact oracle check on a sample."""
    from oracle.oracle import Oracle, SQ_IEEE
    cfg = c2_config()
    B, N, A, K = 4096, 64, 32, 20
    rng = np.random.default_rng(77)
    x0 = rng.integers(0, 2000, size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    env = make_env(cfg, B)
    env.reset_topology(x0, 0.0, v0)
    sample = np.sort(rng.choice(B, size=48, replace=False))
    orc = Oracle(cfg, batch=len(sample), sq_mode=SQ_IEEE, threads=8)
    orc.reset(x0[sample], np.zeros((len(sample), N)), v0[sample])
    T = 30
    for t in range(T):
        acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, done = env.step(acts, t)
        torch.cuda.synchronize()
        a_t = torch.as_tensor(acts, device="cuda").long()
        # (1) one-hot section == actions
        assert torch.equal(obs[..., :A].argmax(-1), a_t) and torch.all(obs[..., :A].sum(-1) == 1)
        # (2) histogram rows sum to 1 (or are all zero)
        hs = obs[..., A:].sum(-1)
        assert torch.all(((hs - 1).abs() < 1e-12) | (hs == 0))
        # (3) rewards against collision counts recomputed in torch (reward_design 2)
        cnt = torch.zeros((B, A), dtype=torch.long, device="cuda").scatter_add_(1, a_t, torch.ones_like(a_t))
        c = cnt.gather(1, a_t)
        assert torch.all(rew[c == 1] == 1)
        assert torch.all(rew[c > 2] == -c[c > 2].double())
        assert torch.all((rew[c == 2] == 0) | (rew[c == 2] == -2))
        # (4) sampled envs bit-exact vs the oracle
        o_rew, o_chobs = orc.step(STEP_MY_STEP, acts[sample], t)
        o_state = orc.obtain_state(acts[sample], o_chobs, o_rew)
        assert np.array_equal(obs[sample].cpu().numpy(), o_state), t
        assert np.array_equal(rew[sample].cpu().numpy(), o_rew), t
    # (5) positions stay in [0, L) and sequence numbers equal the slot count on the diagonal
    st = env.export_state()
    assert st["pos_x"].min() >= 0 and st["pos_x"].max() < 2000
    diag = torch.diagonal(st["seq"], dim1=1, dim2=2)
    assert torch.all(diag == T)
    assert torch.all(st["seq"] <= T)
    env.check()


def test_replica_envs_are_identical_at_scale():
    """Idempotence across the batch: 2048 copies of one env stay identical."""
    cfg = bench_config(128, 64, 4000.0)
    B, N, A = 2048, 128, 64
    rng = np.random.default_rng(5)
    x0 = rng.integers(0, 4000, size=N).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=N)
    env = make_env(cfg, B, dtype=torch.float32)
    env.reset_topology(x0, 0.0, v0)
    for t in range(12):
        a = rng.integers(0, A, size=N).astype(np.int32)
        obs, rew, _ = env.step(a, t)
    torch.cuda.synchronize()
    assert torch.all(obs == obs[0:1]) and torch.all(rew == rew[0:1])
    st = env.export_state()
    for k in ("seq", "age", "x"):
        assert torch.all(st[k] == st[k][0:1])


@pytest.mark.parametrize("N,A,K,rd,toy", [(64, 32, 20, 2, False), (64, 32, 20, 1, False), (64, 32, 40, 5, False),
                                           (64, 32, 20, 3, False), (64, 32, 20, 4, False),
                                           (4, 3, 20, 2, True), (4, 3, 10, 1, True), (63, 31, 10, 2, False),
                                           (33, 7, 20, 2, False), (64, 5, 8, 2, False), (1, 1, 4, 2, False),
                                           (17, 32, 64, 2, False), (64, 64, 20, 2, False), (48, 40, 10, 1, False),
                                           (64, 33, 20, 5, False)])
def test_fast64_kernel_matches_general_kernel_and_oracle(N, A, K, rd, toy):
    """The headline-config kernel (csrc/step_fast64.hpp, f32 and f64 outputs,
    default State flags) against the general kernel (f64 outputs, forced with
    DIRAL_OPT_KERNEL_PATH) and the oracle: f64 states identical, f32 states equal to
    their cast, rewards, positions and every table plane identical."""
    from oracle.oracle import Oracle, SQ_IEEE

    L = 100.0 if toy else 30.0 * N + 100
    cfg = bench_config(N, A, L, reward_design=rd, congestion_test=toy, State=dict(num_bins=K),
                       communication_range=250.0 if N > 8 else 40.0)
    rng = np.random.default_rng(1000 + N + A + K + rd)
    B = 40
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    y0 = rng.integers(0, 2, size=(B, N)).astype(np.float64) if toy else np.zeros((B, N))
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    fast, fast64, gen = (make_env(cfg, B, dtype=torch.float32), make_env(cfg, B, dtype=torch.float64),
                         make_env(cfg, B, dtype=torch.float64))
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    for e in (fast, fast64, gen):
        e.reset_topology(x0, y0, v0)
    orc.reset(x0, y0, v0)
    for t in range(45):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        of, rf, df = fast.step(a, t)
        o6, r6, d6 = fast64.step(a, t)
        with _general_kernel(gen):
            og, rg, dg = gen.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        o_state = orc.obtain_state(a, o_chobs, o_rew)
        torch.cuda.synchronize()
        assert torch.equal(o6, og) and torch.equal(r6, rg) and torch.equal(d6, dg), t
        assert np.array_equal(o6.cpu().numpy(), o_state), t
        assert torch.equal(of, og.to(torch.float32)), t
        assert torch.equal(df, dg)
        assert np.array_equal(of.cpu().numpy(), o_state.astype(np.float32)), t
        if rd in (3, 4):
            assert torch.allclose(rf.double(), rg, rtol=0, atol=1e-6)
            assert np.all(np.abs(r6.cpu().numpy() - o_rew) <= EXP_ATOL), t
        else:
            assert torch.equal(rf, rg.to(torch.float32)), t
            assert np.array_equal(rf.cpu().numpy(), o_rew.astype(np.float32)), t
            assert np.array_equal(r6.cpu().numpy(), o_rew), t
    sf, s6, sg, oe = fast.export_state(), fast64.export_state(), gen.export_state(), orc.export()
    for k in ("pos_x", "vel", "seq", "age", "x", "y"):
        assert torch.equal(sf[k], sg[k]), k
        assert torch.equal(s6[k], sg[k]), k
    assert np.array_equal(sf["seq"].cpu().numpy(), oe["seq"])
    assert np.array_equal(sf["x"].cpu().numpy(), oe["x"])
    assert np.array_equal(sf["pos_x"].cpu().numpy(), oe["pos_x"])
    mf, mg = fast.metrics().cpu().numpy(), gen.metrics().cpu().numpy()
    assert np.array_equal(mf[:, [0, 2, 3]], mg[:, [0, 2, 3]])
    assert np.allclose(mf[:, 1], mg[:, 1], rtol=1e-12, atol=1e-9)
    fast.check()
    gen.check()


@pytest.mark.parametrize("margin", ["0", "2500", "16384"])
@pytest.mark.parametrize("N,A,K,grid", [(64, 32, 20, False), (64, 32, 20, True), (61, 16, 33, False)])
def test_fast64_float32_screening_of_the_bin_is_exact(N, A, K, grid, margin, monkeypatch):
    """The fast quads of csrc/step_fast64.hpp take the histogram bin from float32
    copies unless its fraction lies in a band around an integer, and from float64
    inside it.  DIRAL_F32_MARGIN (read at create) switches the screening off (0) or
    widens the band (2500 / 65536 of a bin: 8 % of the entries redone in float64;
    16384: half of them), so that both bodies and their hand-over run on every step;
    `grid`: positions and speeds on the grid of bin edges - every difference an exact
    edge hit.  States, rewards and tables against the oracle, bit for bit."""
    from oracle.oracle import Oracle, SQ_IEEE

    L = 30.0 * N + 100
    cfg = bench_config(N, A, L, State=dict(num_bins=K))
    rng = np.random.default_rng(77 + N + K + int(margin))
    B = 24
    if grid:
        x0 = 25.0 * rng.integers(0, int(L) // 25, size=(B, N)).astype(np.float64)
        v0 = 25.0 * rng.integers(0, 3, size=(B, N)).astype(np.float64)
    else:
        x0 = rng.uniform(0, L, size=(B, N))
        v0 = rng.uniform(1.1, 2.7, size=(B, N))
    y0 = np.zeros((B, N))
    monkeypatch.setenv("DIRAL_F32_MARGIN", margin)
    env = make_env(cfg, B, dtype=torch.float64)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    env.reset_topology(x0, y0, v0)
    orc.reset(x0, y0, v0)
    for t in range(30):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        o, r, d = env.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        o_state = orc.obtain_state(a, o_chobs, o_rew)
        assert np.array_equal(o.cpu().numpy(), o_state), t
        assert np.array_equal(r.cpu().numpy(), o_rew), t
    assert (env.last_kernel() & 15) == _fam(N)
    se, oe = env.export_state(), orc.export()
    for k in ("seq", "x", "pos_x"):
        assert np.array_equal(se[k].cpu().numpy(), oe[k]), k
    env.check()


@pytest.mark.parametrize("name", ["g1_step_rd2", "g1_ch_rd2", "g1_design", "g1_flags_all"])
def test_testenv_shim_on_gpu_reads_like_the_reference(name):
    """`from diral_amd import TestEnv` driven with the reference's own call
    sequence (main_test.py:89-164) returns the reference's Python shapes and the
    recorded values."""
    from diral_amd import TestEnv
    g = Golden(name)
    env = TestEnv(**g.cfg_dict)
    env.reset_mobility_env()
    step = {STEP_MY_STEP: env.my_step, STEP_MY_STEP_CH: env.my_step_ch, STEP_DESIGN: env.my_step_design}
    for i, mode, acts, t, (ep, eps) in g.steps():
        obs, rews = step[mode](acts, t)
        state = env.obtain_state(obs, acts, list(rews), ep, eps)
        assert isinstance(obs, dict) and isinstance(state, list) and len(state) == 4
        assert np.array_equal(rews, g["rews"][i])
        assert ulp_diff(np.array([obs[u] for u in range(4)]), g["chobs"][i]) <= DIST_ULP
        assert ulp_diff(np.array(state), g["state"][i]) <= DIST_ULP
        assert env.get_x_pos() == list(g["pos_x"][i])
        assert env.network.get_information_age(t) == list(g["ia"][i])


@pytest.mark.parametrize("stale_frac", [0.0, 0.02, 0.5])
def test_fast64_packed_merge_and_its_fallback_on_stale_tables(stale_frac):
    """The fast kernel merges 16-bit (rank, source) keys when every entry of a
    wave is younger than 1023 slots (or never heard) and falls back to 32-bit
    keys otherwise.  Imported tables with arbitrarily stale sequence numbers
    exercise both, per wave, against the oracle."""
    from oracle.oracle import Oracle, SQ_IEEE
    cfg = c2_config().replace(track_arrival=False)
    B, N, A, T0 = 12, 64, 32, 5000
    rng = np.random.default_rng(int(stale_frac * 1000) + 3)
    x0 = rng.integers(0, 2000, size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    seq = T0 - rng.integers(0, 40, size=(B, N, N))                      # fresh-ish entries
    stale = rng.random((B, N, N)) < stale_frac
    seq = np.where(stale, rng.integers(0, T0 - 1023, size=(B, N, N)), seq)  # lag >= 1023, some seq == 0
    seq[:, np.arange(N), np.arange(N)] = T0
    age = rng.integers(0, 40, size=(B, N, N))
    # a table entry is the subject's stamp at that sequence number, so entries about the same
    # subject with equal seq carry equal xpos in every reachable state; keep that invariant
    kk = np.arange(N)[None, None, :]
    bb = np.arange(B)[:, None, None]
    x = ((kk * 7919 + seq * 104729 + bb * 31) % 200000) / 100.0
    fast = make_env(cfg, B, dtype=torch.float32)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    fast.reset_topology(x0, None, v0)
    fast.import_state(seq=seq, age=age, x=x)
    orc.reset(x0, np.zeros((B, N)), v0)
    orc.import_state(seq=seq, age=age, x=x, y=np.zeros((B, N, N)))
    for t in range(30):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, _ = fast.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        o_state = orc.obtain_state(a, o_chobs, o_rew)
        torch.cuda.synchronize()
        assert np.array_equal(obs.cpu().numpy(), o_state.astype(np.float32)), t
    st, oe = fast.export_state(), orc.export()
    assert np.array_equal(st["seq"].cpu().numpy(), oe["seq"])
    assert np.array_equal(st["age"].cpu().numpy(), np.minimum(oe["age"], 255))
    assert np.array_equal(st["x"].cpu().numpy(), oe["x"])
    fast.check()


@pytest.mark.parametrize("N,A,full,ptype", [(64, 32, True, 2), (64, 32, False, 1), (64, 32, True, 1),
                                            (70, 9, True, 1), (130, 40, True, 2), (200, 16, False, 1),
                                            (5, 3, True, 1)])
def test_secondary_observation_modes_vs_oracle(N, A, full, ptype):
    """add_positional_dist (sorted signed true distances / max)
    and add_positional_dist_type 1 (inf-norm scaled, weighted np.histogram with
    explicit edges = sequential prefix sums), bit-exact against the oracle."""
    cfg = bench_config(N, A, 30.0 * N + 50, State=dict(add_positional_dist=full, add_positional_dist_type=ptype,
                                                        num_bins=20 if N != 70 else 7))
    random_rollout(cfg, B=5, T=26, seed=500 + N + ptype)


@pytest.mark.parametrize("name", ["s1_sps_int_threshold", "s2_sps_frac_threshold", "s3_sps_small_window", "s4_sps_window64",
                                  "s5_sps_window100", "s6_sps_window200", "s7_sps_window300"])
def test_sps_policy_matches_the_reference_fixtures(name):
    """the SPS baseline on the device against fixtures recorded from the
    reference's own algorithms/v2x_sps.py (tests/golden/gen_golden.py `sps`: one
    SemiPersistentScheduling object per agent, its random.randint / random.random /
    random.choice calls mocked with the recorded draws): action, reselection counter and
    prev_action after every step, incl. stable-sort ties, repeated 3 dB threshold raises
    and a non-integer threshold; s4-s7: windows of 64 / 100 / 200 subframes (1, 2, 4 per lane of the
    wave-cooperative kernel) and 300 (one thread per agent)."""
    import os
    from diral_amd.sps import SpsPolicy
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    A = int(g["A"])
    T, n = g["actions"].shape
    assert T * n >= 200 and int(g["reselections"]) >= 40
    pol = SpsPolicy(1, n, A, rssi_threshold=float(g["threshold"]), seed=5)
    pol.prev_action.copy_(torch.as_tensor(g["init_prev"]).view(1, n))
    pol.counter.copy_(torch.as_tensor(g["init_counter"]).view(1, n))
    step = float(g["tie_step"])
    for t in range(T):
        win = torch.as_tensor(g["codes"][t].astype(np.float64) * step).view(1, n, A)
        got = pol.step(win, g["draw_counter"][t], g["draw_keep"][t], g["draw_choice"][t]).cpu().numpy()[0]
        assert np.array_equal(got, g["actions"][t]), (t, np.argwhere(got != g["actions"][t])[:4])
        assert np.array_equal(pol.counter.cpu().numpy()[0], g["counters"][t]), t
        assert np.array_equal(pol.prev_action.cpu().numpy()[0], g["prev_actions"][t]), t


def test_sps_device_draws_follow_the_reference_distributions():
    """The un-injected branches of diral_sps_init / diral_sps_step (device RNG): initial
    prev_action in [0, window], counters in [5, 15], new counters in [5, 16], keep
    probability 0.8 (v2x_sps.py:14-15, 91-93)."""
    from diral_amd.sps import SpsPolicy
    B, N, A = 64, 64, 16
    pol = SpsPolicy(B, N, A, rssi_threshold=-110.0, seed=11)
    pa, cn = pol.prev_action.cpu().numpy(), pol.counter.cpu().numpy()
    assert pa.min() == 0 and pa.max() == A - 1 and cn.min() == 5 and cn.max() == 15
    assert abs(cn.mean() - 10.0) < 0.2
    win = torch.full((B, N, A), -150.0, dtype=torch.float64, device="cuda:0")   # everything free
    resel = expired = 0
    seen = set()
    for t in range(40):
        before_c, before_a = pol.counter.clone(), pol.prev_action.clone()
        pol.step(win)
        exp = (before_c == 0)
        expired += int(exp.sum())
        resel += int((exp & (pol.prev_action != before_a)).sum())
        seen |= set(pol.counter[exp].cpu().numpy().tolist())
    assert expired > 5000 and seen == set(range(5, 17))
    # a re-selection picks among max(1, ceil(min(A/5, len(sA)))) = 4 best of 15 others: it differs from
    # the previous action always (prev is excluded), so the changed fraction IS the 20 %
    assert abs(resel / expired - 0.2) < 0.03, resel / expired


@pytest.mark.parametrize("dtype,N,A", [(torch.float32, 64, 32), (torch.float64, 64, 32), (torch.float32, 24, 100),
                                       (torch.float64, 20, 160)])
def test_sps_from_channel_obs_fused_equals_two_step_path(dtype, N, A):
    """`diral_sps_step_chobs` (window + decision in one launch, window built only by re-selecting
    agents) == `diral_sps_window_from_chobs` + `diral_sps_step` with the same draws, decision for
    decision; and the device window is the documented formula (rssi_from_channel_obs in torch)."""
    from diral_amd.sps import SpsPolicy, rssi_from_channel_obs
    cfg = bench_config(N, A, 2000.0)
    B = 16
    env = make_env(cfg, B, dtype=dtype)
    env.reset_topology(seed=12)
    fused, two = SpsPolicy(B, N, A, seed=4), SpsPolicy(B, N, A, seed=4)
    assert torch.equal(fused.prev_action, two.prev_action) and torch.equal(fused.counter, two.counter)
    acts = fused.prev_action.clone()
    rng = np.random.default_rng(5)
    nres = 0
    for t in range(120):
        chobs, _ = env.my_step(acts, t)
        win = two.window_from_chobs(chobs, acts)
        ref = rssi_from_channel_obs(chobs, acts)
        assert torch.allclose(win, ref, rtol=0, atol=1e-9), t
        assert torch.equal(win == -60.0, ref == -60.0) and torch.equal(win == -200.0, ref == -200.0)
        dc = rng.integers(5, 17, size=(B, N)).astype(np.int32)
        dk = rng.random((B, N))
        dch = rng.integers(0, 1 << 20, size=(B, N)).astype(np.int32)
        a1 = fused.step_from_chobs(chobs, acts, dc, dk, dch)
        a2 = two.step(win, dc, dk, dch)
        assert torch.equal(a1, a2), t
        assert torch.equal(fused.counter, two.counter) and torch.equal(fused.prev_action, two.prev_action), t
        nres += int((a1 != acts).sum())
        acts = a1
    assert nres > 3 * N                                  # re-selections really happened
    env.check()


def test_sps_policy_drives_the_env():
    """SPS closes the loop on the device: env channel obs -> RSSI-like window ->
    actions, no host round trip; collisions drop well below the uniform-random level."""
    from diral_amd.sps import SpsPolicy, rssi_from_channel_obs
    cfg = c2_config()
    B = 64
    env = make_env(cfg, B, dtype=torch.float32)
    env.reset_topology(seed=9)
    pol = SpsPolicy(B, 64, 32, rssi_threshold=-110.0, seed=3)
    acts = pol.prev_action.clone()
    for t in range(150):
        env._step(STEP_MY_STEP, acts, t, want_chobs=True)
        acts = pol.step_from_chobs(env._chobs, acts)
        if t == 99:
            env.metrics(clear=True)
    m = env.metrics().sum(0)
    torch.cuda.synchronize()
    coll = (m[3] / (m[2] + m[3])).item()
    assert coll < 0.80, coll                              # iid-uniform actions give ~0.865
    env.check()


def test_per_env_trace_replay_vs_oracle():
    """load_positions with one trace PER env ([B, T, N], build extension of the
    reference's single [T, N] file): each env against its own oracle."""
    from oracle.oracle import Oracle, SQ_IEEE
    cfg = bench_config(20, 6, 800.0, communication_range=200.0)
    B, N, A, T = 4, 20, 6, 9
    rng = np.random.default_rng(33)
    x0 = rng.integers(0, 800, size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    traces = rng.uniform(0, 800, size=(B, T, N))
    env = make_env(cfg, B)
    env.reset_topology(x0, None, v0)
    env.load_saved_positions(traces)
    orcs = [Oracle(cfg, batch=1, sq_mode=SQ_IEEE) for _ in range(B)]
    for b, o in enumerate(orcs):
        o.reset(x0[b:b + 1], np.zeros((1, N)), v0[b:b + 1])
        o.set_trace(traces[b])
    for t in list(range(14)) + [40, 41, 5]:
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, chobs, _ = gpu_step(env, STEP_MY_STEP, a, t)
        for b, o in enumerate(orcs):
            o_rew, o_chobs = o.step(STEP_MY_STEP, a[b:b + 1], t)
            assert np.array_equal(obs[b], o.obtain_state(a[b:b + 1], o_chobs, o_rew)[0]), (t, b)
            assert np.array_equal(rew[b], o_rew[0])
    px = env.get_x_pos().cpu().numpy()
    assert np.array_equal(px, traces[:, 5 % T, :])
    env.load_saved_positions(None)            # back to the velocity model
    env.step(np.zeros((B, N), np.int32), 0)
    torch.cuda.synchronize()
    assert not np.array_equal(env.get_x_pos().cpu().numpy(), px)


def test_long_run_past_the_packed_rank_horizon():
    """1300 slots in a sparse topology: pairs that never hear each other keep
    seq == 0 entries whose lag passes 1023 (the packed-merge rank saturates) and
    ages saturate at 255; everything must still match the oracle bit for bit."""
    from oracle.oracle import Oracle, SQ_IEEE
    cfg = bench_config(24, 6, 3000.0, communication_range=60.0)
    B, N, A = 3, 24, 6
    rng = np.random.default_rng(8)
    x0 = rng.integers(0, 3000, size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    fast = make_env(cfg, B, dtype=torch.float32)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE)
    fast.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    for t in range(1300):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, _ = fast.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        if t % 50 == 49 or t > 1270:
            o_state = orc.obtain_state(a, o_chobs, o_rew)
            torch.cuda.synchronize()
            assert np.array_equal(obs.cpu().numpy(), o_state.astype(np.float32)), t
    st, oe = fast.export_state(), orc.export()
    assert (oe["seq"] == 0).any() and (oe["age"] > 255).any()
    assert np.array_equal(st["seq"].cpu().numpy(), oe["seq"])
    assert np.array_equal(st["age"].cpu().numpy(), np.minimum(oe["age"], 255))
    assert np.array_equal(st["x"].cpu().numpy(), oe["x"])
    fast.check()


def test_step_is_capturable_in_a_hip_graph():
    """diral_env_step only enqueues on the given stream (no allocation, no host
    sync): a captured sequence of steps replays to the same result as eager."""
    cfg = c2_config()
    B = 32
    rng = np.random.default_rng(4)
    x0 = rng.integers(0, 2000, size=(B, 64)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, 64))
    acts = [torch.as_tensor(rng.integers(0, 32, size=(B, 64)).astype(np.int32), device="cuda") for _ in range(4)]
    eager, graphed = make_env(cfg, B, dtype=torch.float32), make_env(cfg, B, dtype=torch.float32)
    for e in (eager, graphed):
        e.reset_topology(x0, None, v0)
    for i in range(4):
        obs_e, rew_e, _ = eager.step(acts[i], i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(4):
                obs_g, rew_g, _ = graphed.step(acts[i], i)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(obs_e, obs_g) and torch.equal(rew_e, rew_g)
    se, sg = eager.export_state(), graphed.export_state()
    assert torch.equal(se["seq"], sg["seq"]) and torch.equal(se["x"], sg["x"])


def test_cabi_optional_outputs_and_seeds():
    """C-ABI details: NULL reward/done/chobs outputs are allowed; the device
    topology RNG is a pure function of the seed; metrics clear on request."""
    import ctypes
    from diral_amd import _lib
    cfg = c2_config()
    lib = _lib.load()
    a, b, c = make_env(cfg, 16, dtype=torch.float32), make_env(cfg, 16, dtype=torch.float32), \
        make_env(cfg, 16, dtype=torch.float32)
    a.reset_topology(seed=11)
    b.reset_topology(seed=11)
    c.reset_topology(seed=12)
    sa, sb, sc = a.export_state(False), b.export_state(False), c.export_state(False)
    assert torch.equal(sa["pos_x"], sb["pos_x"]) and torch.equal(sa["vel"], sb["vel"])
    assert not torch.equal(sa["pos_x"], sc["pos_x"])
    acts = a.sample(seed=5)
    assert torch.equal(acts, b.sample(seed=5)) and not torch.equal(acts, b.sample(seed=6))
    # state only: no reward / done / channel-obs buffers
    obs = torch.zeros((16, 64, 52), dtype=torch.float32, device="cuda")
    st = lib.diral_env_step(a._h, 0, acts.data_ptr(), 0, obs.data_ptr(), None, None, None, 0, 0.0, 1.0,
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    ob, _, _ = b.step(acts, 0)
    torch.cuda.synchronize()
    assert torch.equal(obs, ob)
    # nothing at all requested but the state advance (state_out NULL => general kernel)
    assert lib.diral_env_step(a._h, 0, acts.data_ptr(), 1, None, None, None, None, 0, 0.0, 1.0, None) == 0
    b.step(acts, 1)
    torch.cuda.synchronize()
    assert torch.equal(a.export_state()["seq"], b.export_state()["seq"])
    # bad arguments are rejected, not executed
    assert lib.diral_env_step(a._h, 7, acts.data_ptr(), 0, None, None, None, None, 0, 0.0, 1.0, None) == -1
    assert lib.diral_env_step(a._h, 0, None, 0, None, None, None, None, 0, 0.0, 1.0, None) == -1
    assert lib.diral_env_step(a._h, 0, acts.data_ptr(), 0, None, None, None, None, 5, 0.0, 1.0, None) == -1
    m = b.metrics(clear=True)
    assert torch.all(m[:, 0] == 2) and torch.all(b.metrics()[:, 0] == 0)
    assert b.hbm_bytes() > 16 * 64 * 64 * 12


def test_two_envs_on_two_streams():
    """Handles are independent: two envs stepped concurrently on different HIP
    streams give the results of stepping them one after the other."""
    cfg = c2_config()
    B = 256
    rng = np.random.default_rng(21)
    acts = [torch.as_tensor(rng.integers(0, 32, size=(B, 64)).astype(np.int32), device="cuda") for _ in range(6)]
    ser = [make_env(cfg, B, dtype=torch.float32) for _ in range(2)]
    par = [make_env(cfg, B, dtype=torch.float32) for _ in range(2)]
    for i in range(2):
        ser[i].reset_topology(seed=50 + i)
        par[i].reset_topology(seed=50 + i)
    for t in range(6):
        for i in range(2):
            ser[i].step(acts[(t + i) % 6], t)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for t in range(6):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                par[i].step(acts[(t + i) % 6], t)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(ser[i]._obs, par[i]._obs)
        assert torch.equal(ser[i].export_state()["x"], par[i].export_state()["x"])


def test_example_rollout_script_runs():
    import subprocess
    import sys
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    for pol in ("sps", "random"):
        out = subprocess.run([sys.executable, "examples/rollout_sps.py", "--envs", "64", "--slots", "60",
                              "--policy", pol], cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "collision fraction" in out.stdout


def _general_kernel(env):
    """Context manager: this env's steps run on the general step_kernel
    (DIRAL_OPT_KERNEL_PATH, include/diral_env.h)."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        env.force_general_kernel(True)
        try:
            yield
        finally:
            env.force_general_kernel(False)
    return cm()


@pytest.mark.parametrize("N,A,K,rd,toy", [(256, 64, 20, 2, False), (128, 64, 20, 2, False), (130, 33, 10, 1, False),
                                           (65, 3, 8, 2, False), (200, 64, 40, 5, False), (256, 64, 20, 3, False),
                                           (100, 7, 20, 4, False), (129, 64, 21, 2, False), (255, 1, 20, 2, False),
                                           (192, 48, 12, 2, True)])
def test_wide_kernel_matches_general_kernel_and_oracle(N, A, K, rd, toy):
    """csrc/step_wide.hpp (64 < N <= 256, default State flags, all y == 0; 8-bit rank
    merge, rank-indexed xpos hand-over) against the general kernel and the oracle:
    f64 states identical, f32 states equal to their cast, rewards, positions, metrics
    and every table plane identical."""
    from oracle.oracle import Oracle, SQ_IEEE
    L = 15.0 * N + 100
    cfg = bench_config(N, A, L, reward_design=rd, congestion_test=toy, State=dict(num_bins=K))
    rng = np.random.default_rng(2000 + N + A + K + rd)
    B = 6
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    w32, w64, gen = (make_env(cfg, B, dtype=torch.float32), make_env(cfg, B, dtype=torch.float64),
                     make_env(cfg, B, dtype=torch.float64))
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    for e in (w32, w64, gen):
        e.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    for t in range(30):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        o3, r3, d3 = w32.step(a, t)
        o6, r6, d6 = w64.step(a, t)
        with _general_kernel(gen):
            og, rg, dg = gen.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        o_state = orc.obtain_state(a, o_chobs, o_rew)
        torch.cuda.synchronize()
        assert torch.equal(o6, og) and torch.equal(d6, dg) and torch.equal(d3, dg), t
        assert np.array_equal(o6.cpu().numpy(), o_state), t
        assert torch.equal(o3, og.to(torch.float32)), t
        if rd in (3, 4):
            assert np.all(np.abs(r6.cpu().numpy() - o_rew) <= EXP_ATOL), t
            assert torch.allclose(r3.double(), rg, rtol=0, atol=1e-6)
        else:
            assert torch.equal(r6, rg) and torch.equal(r3, rg.to(torch.float32)), t
            assert np.array_equal(r6.cpu().numpy(), o_rew), t
    s3, s6, sg, oe = w32.export_state(), w64.export_state(), gen.export_state(), orc.export()
    for k in ("pos_x", "vel", "seq", "age", "x", "y"):
        assert torch.equal(s3[k], sg[k]), k
        assert torch.equal(s6[k], sg[k]), k
    assert np.array_equal(s6["seq"].cpu().numpy(), oe["seq"])
    assert np.array_equal(s6["x"].cpu().numpy(), oe["x"])
    assert np.array_equal(s6["age"].cpu().numpy(), np.minimum(oe["age"], 255))
    m6, mg = w64.metrics().cpu().numpy(), gen.metrics().cpu().numpy()
    assert np.array_equal(m6[:, [0, 2, 3]], mg[:, [0, 2, 3]])
    assert np.allclose(m6[:, 1], mg[:, 1], rtol=1e-12, atol=1e-9)
    for e in (w32, w64, gen):
        e.check()


@pytest.mark.parametrize("N,stale_frac,lag_hi", [(256, 0.0, 40), (256, 0.01, 40), (128, 0.3, 40), (150, 0.05, 40),
                                                 (256, 0.0, 8), (128, 0.0, 8), (150, 0.0, 5), (256, 0.0, -1), (128, 0.0, -1),
                                                 (200, 0.002, -1), (256, 0.0, 9),
                                                 (128, -0.3, 40), (256, -0.2, 40), (150, -0.4, -1), (100, -0.5, 8)])
def test_wide_rank_merge_and_its_fallback_on_stale_tables(N, stale_frac, lag_hi):
    """step_wide merges a pass of subject columns as 8-level thermometer codes when every entry
    of the pass is at most 7 slots behind (or never heard), as 8-bit ranks when younger than 255
    slots, and takes the 32-bit (seq, source) path otherwise; imported tables with chosen lags
    exercise all three against the oracle (lag_hi -1: subjects alternate between lags < 8 and
    < 40, so consecutive passes of one launch take different paths; 9: a single lag-8 entry
    here and there is enough to leave the thermometer path).  A negative stale_frac: the stale entries of a subject
    all hold ONE sequence number (what a highway that broke into clusters leaves behind) - except for every seventh
    subject, which holds two: the byte ranks carry the former (they share the lowest rank), the latter still takes the
    32-bit path."""
    from oracle.oracle import Oracle, SQ_IEEE
    A, T0, B = 64, 5000, 4
    cfg = bench_config(N, A, 15.0 * N + 100)
    rng = np.random.default_rng(int(abs(stale_frac) * 1000) + N + 7 * lag_hi + (500 if stale_frac < 0 else 0))
    x0 = rng.integers(0, int(cfg.highway_length), size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    if lag_hi > 0:
        hi = np.full((B, N, 1), lag_hi)
    else:
        hi = np.where((np.arange(N) // 4) % 3 == 0, 40, 8)[None, :, None] * np.ones((B, 1, 1), dtype=np.int64)
    seq = T0 - rng.integers(0, hi, size=(B, N, N))             # [env][subject][viewer]
    stale = rng.random((B, N, N)) < abs(stale_frac)
    if stale_frac < 0:
        common = rng.integers(1, T0 - 255, size=(B, N, 1)) + np.zeros((1, 1, N), dtype=np.int64)
        two = (np.arange(N) % 7 == 3)[None, :, None] & (rng.random((B, N, N)) < 0.5)
        seq = np.where(stale, np.where(two, np.maximum(common - 3, 1), common), seq)
    else:
        seq = np.where(stale, rng.integers(0, T0 - 255, size=(B, N, N)), seq)   # lag >= 255, some seq == 0
    seq = np.where(rng.random((B, N, N)) < 0.05, 0, seq)                    # never-heard entries
    seq[:, np.arange(N), np.arange(N)] = T0
    age = rng.integers(0, 40, size=(B, N, N))
    kk = np.arange(N)[None, None, :]
    bb = np.arange(B)[:, None, None]
    x = ((kk * 7919 + seq * 104729 + bb * 31) % 200000) / 100.0   # xpos is a function of (subject, seq)
    env = make_env(cfg, B, dtype=torch.float64)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    env.reset_topology(x0, None, v0)
    env.import_state(seq=seq, age=age, x=x)
    orc.reset(x0, np.zeros((B, N)), v0)
    orc.import_state(seq=seq, age=age, x=x, y=np.zeros((B, N, N)))
    for t in range(12):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, _ = env.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        o_state = orc.obtain_state(a, o_chobs, o_rew)
        torch.cuda.synchronize()
        assert np.array_equal(obs.cpu().numpy(), o_state), t
    st, oe = env.export_state(), orc.export()
    assert np.array_equal(st["seq"].cpu().numpy(), oe["seq"])
    assert np.array_equal(st["age"].cpu().numpy(), np.minimum(oe["age"], 255))
    assert np.array_equal(st["x"].cpu().numpy(), oe["x"])
    env.check()


@pytest.mark.parametrize("N,A,K,rd", [(64, 32, 20, 2), (64, 32, 20, 3), (64, 32, 10, 4), (33, 7, 20, 2), (4, 3, 20, 3),
                                       (64, 64, 20, 2), (50, 48, 20, 4),
                                       (256, 64, 20, 2), (128, 64, 20, 3), (130, 33, 10, 4), (65, 3, 8, 2),
                                       (200, 64, 40, 2)])
def test_fast_paths_run_my_step_ch_like_the_general_kernel_and_the_oracle(N, A, K, rd):
    """my_step_ch (PRR reward, test_env.py:351-443) on the specialised kernels
    (step_fast64 / step_wide, CH instantiation) against the general kernel and the
    oracle: integer reception counts bit-exact, so R and the rd 2 rewards are
    identical; rd 3/4 go through exp (absolute tolerance); states, tables and the
    PRR metric columns identical."""
    from oracle.oracle import Oracle, SQ_IEEE
    L = (30.0 if N <= 64 else 15.0) * N + 100
    cfg = bench_config(N, A, L, reward_design=rd, State=dict(num_bins=K),
                       communication_range=250.0 if N > 8 else 40.0)
    rng = np.random.default_rng(3000 + N + A + K + rd)
    B = 8
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    f32, f64, gen = (make_env(cfg, B, mode="my_step_ch", dtype=torch.float32),
                     make_env(cfg, B, mode="my_step_ch", dtype=torch.float64),
                     make_env(cfg, B, mode="my_step_ch", dtype=torch.float64))
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    for e in (f32, f64, gen):
        e.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    for t in range(30):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        if t % 3 == 2:                                       # some sticky slots: fewer collisions, R spread
            a[:, ::2] = prev[:, ::2]
        prev = a
        o3, r3, _ = f32.step(a, t)
        o6, r6, d6 = f64.step(a, t)
        with _general_kernel(gen):
            og, rg, dg = gen.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP_CH, a, t)
        o_state = orc.obtain_state(a, o_chobs, o_rew)
        torch.cuda.synchronize()
        assert torch.equal(o6, og) and torch.equal(d6, dg), t
        assert np.array_equal(o6.cpu().numpy(), o_state), t
        assert torch.equal(o3, og.to(torch.float32)), t
        if rd == 2:
            assert torch.equal(r6, rg) and torch.equal(r3, rg.to(torch.float32)), t
            assert np.array_equal(r6.cpu().numpy(), o_rew), t
        else:
            assert torch.equal(r6, rg), t                    # same device exp() on the same R
            assert np.all(np.abs(r6.cpu().numpy() - o_rew) <= EXP_ATOL), t
            assert torch.allclose(r3.double(), rg, rtol=0, atol=1e-6)
    s6, sg, oe = f64.export_state(), gen.export_state(), orc.export()
    for k in ("pos_x", "seq", "age", "x"):
        assert torch.equal(s6[k], sg[k]), k
    assert np.array_equal(s6["seq"].cpu().numpy(), oe["seq"])
    m6, mg, mo = f64.metrics().cpu().numpy(), gen.metrics().cpu().numpy(), orc.metrics()
    assert np.array_equal(m6[:, [0, 2, 3, 5]], mg[:, [0, 2, 3, 5]])
    assert np.array_equal(m6[:, 4], mg[:, 4])                # PRR sums: same values, same summation order
    assert np.allclose(m6[:, 4] / m6[:, 5], mo[:, 4] / mo[:, 5], rtol=0, atol=1e-12)
    for e in (f32, f64, gen):
        e.check()


def test_wide_long_run_in_and_out_of_the_rank_window():
    """700 slots of a sparse 96-vehicle highway: pairs meet and drift apart, so entries
    with a real sequence number age past the 8-bit rank window (lag >= 255) and come
    back - step_wide alternates between its rank path and its 32-bit path per pass and
    must match the oracle bit for bit throughout."""
    from oracle.oracle import Oracle, SQ_IEEE
    N, A, L, B = 96, 8, 4000.0, 3
    cfg = bench_config(N, A, L, communication_range=100.0)
    rng = np.random.default_rng(18)
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    env = make_env(cfg, B, dtype=torch.float64)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=4)
    env.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    for t in range(700):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, _ = env.step(a, t)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        if t % 25 == 24 or t > 680:
            o_state = orc.obtain_state(a, o_chobs, o_rew)
            torch.cuda.synchronize()
            assert np.array_equal(obs.cpu().numpy(), o_state), t
            assert np.array_equal(rew.cpu().numpy(), o_rew), t
    st, oe = env.export_state(), orc.export()
    seq = oe["seq"]
    own = seq[:, np.arange(N), np.arange(N)]                  # [B, k]: the subject's own sequence number
    lag = own[:, None, :] - seq                               # seq is [B, viewer, subject]
    assert ((lag >= 255) & (seq > 0)).any(), "the run never left the rank window"
    assert ((lag < 255) & (seq > 0)).any()
    assert np.array_equal(st["seq"].cpu().numpy(), seq)
    assert np.array_equal(st["age"].cpu().numpy(), np.minimum(oe["age"], 255))
    assert np.array_equal(st["x"].cpu().numpy(), oe["x"])
    env.check()


@pytest.mark.parametrize("N,A,L,form", [(128, 64, 4000.0, None), (128, 16, 4000.0, "packed"), (200, 32, 5000.0, "packed")])
def test_packed_form_long_run_on_highways_that_break_apart(N, A, L, form, monkeypatch):
    """The packed table form of step_wide on BASELINE configs[4]'s density, 650 slots with the velocities redrawn every 25:
    some highways break into clusters that no longer hear each other, their passes leave the codes (flagged passes that read
    and write the code words themselves, byte ranks), entries about the other cluster age past 254 stamps (the shared lowest
    rank at N <= 128, the 32-bit path with its unpack / repack stages at N > 128) and come back when the clusters meet
    again; those envs ask for the first blocks of the next launch.  Against the oracle: state and reward every 50 slots and
    over the last 30, every table plane at the end."""
    from oracle.oracle import Oracle, SQ_IEEE
    from diral_amd.config import KERNEL_PACKED, KERNEL_WIDE
    if form:
        monkeypatch.setenv("DIRAL_TABLE_FORM", form)
    B, T = 6, 650
    cfg = bench_config(N, A, L, mobility_vary=True)
    rng = np.random.default_rng(N + A)
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    x0[0] = np.concatenate([rng.integers(0, 1200, size=N // 2), rng.integers(2400, 3600, size=N - N // 2)])   # two clusters from the start
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    env = make_env(cfg, B, dtype=torch.float64)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    env.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    for t in range(T):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, _ = env.step(a, t)
        assert env.last_kernel() & (15 | KERNEL_PACKED) == KERNEL_WIDE | KERNEL_PACKED, env.last_kernel()
        o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
        if t % 50 == 49 or t >= T - 30:
            o_state = orc.obtain_state(a, o_chobs, o_rew)
            torch.cuda.synchronize()
            assert np.array_equal(obs.cpu().numpy(), o_state), t
            assert np.array_equal(rew.cpu().numpy(), o_rew), t
        if t % 25 == 24:
            draws = rng.integers(1, 4, size=(B, N)).astype(np.uint8)
            env.update_velocity(draws)
            orc.update_velocity(draws)
    st, oe = env.export_state(), orc.export()
    seq = oe["seq"]
    own = seq[:, np.arange(N), np.arange(N)]
    lag = own[:, None, :] - seq
    if N <= 128:
        assert ((lag >= 254) & (seq > 0)).any(), "no entry ever fell 254 stamps behind"
    assert ((lag >= 8) & (lag < 254) & (seq > 0)).any() and ((lag < 8) & (seq > 0)).any()
    assert np.array_equal(st["seq"].cpu().numpy(), seq)
    assert np.array_equal(st["age"].cpu().numpy(), np.minimum(oe["age"], 255))
    assert np.array_equal(st["x"].cpu().numpy(), oe["x"])
    env.check()


@pytest.mark.parametrize("N,A,K", [(64, 32, 20), (64, 5, 10), (6, 3, 20), (40, 48, 20), (256, 64, 20), (128, 16, 10),
                                   (130, 33, 21), (70, 4, 8)])
def test_specialised_kernels_run_my_step_design_like_the_general_kernel_and_the_oracle(N, A, K):
    """my_step_design (the driver's prefill step, test_env.py:269-349: reward by the
    number of same-resource transmitters within 2 Rc) on step_fast64 / step_wide - a
    runtime switch of the my_step instantiation - against the general kernel and the
    oracle, bit for bit."""
    from oracle.oracle import Oracle, SQ_IEEE
    L = (20.0 if N <= 64 else 10.0) * N + 100
    cfg = bench_config(N, A, L, State=dict(num_bins=K), communication_range=100.0 if N > 8 else 20.0)
    rng = np.random.default_rng(4000 + N + A + K)
    B = 6
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    f32, f64, gen = (make_env(cfg, B, mode="my_step_design", dtype=torch.float32),
                     make_env(cfg, B, mode="my_step_design", dtype=torch.float64),
                     make_env(cfg, B, mode="my_step_design", dtype=torch.float64))
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    for e in (f32, f64, gen):
        e.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    for t in range(24):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        o3, r3, _ = f32.step(a, t)
        o6, r6, d6 = f64.step(a, t)
        with _general_kernel(gen):
            og, rg, dg = gen.step(a, t)
        o_rew, o_chobs = orc.step(STEP_DESIGN, a, t)
        o_state = orc.obtain_state(a, o_chobs, o_rew)
        torch.cuda.synchronize()
        assert torch.equal(o6, og) and torch.equal(r6, rg) and torch.equal(d6, dg), t
        assert np.array_equal(o6.cpu().numpy(), o_state) and np.array_equal(r6.cpu().numpy(), o_rew), t
        assert torch.equal(o3, og.to(torch.float32)) and torch.equal(r3, rg.to(torch.float32)), t
    s6, sg, oe = f64.export_state(), gen.export_state(), orc.export()
    for k in ("pos_x", "seq", "age", "x"):
        assert torch.equal(s6[k], sg[k]), k
    assert np.array_equal(s6["seq"].cpu().numpy(), oe["seq"])
    m6, mg = f64.metrics().cpu().numpy(), gen.metrics().cpu().numpy()
    assert np.array_equal(m6[:, [0, 1, 2, 3]], mg[:, [0, 1, 2, 3]])      # integer-valued rewards: sums exact
    for e in (f32, f64, gen):
        e.check()


@pytest.mark.parametrize("N,A,mode", [(64, 32, STEP_MY_STEP_CH), (64, 32, STEP_MY_STEP), (40, 6, STEP_DESIGN),
                                      (256, 64, STEP_MY_STEP_CH), (130, 20, STEP_MY_STEP_CH), (128, 64, STEP_MY_STEP)])
def test_arrival_stamps_and_information_age_on_the_specialised_kernels(N, A, mode):
    """track_arrival on step_fast64 / step_wide: last_arrival_time (test_env.py:436 stamps,
    network.py:394 '-1' side effect) and get_information_age (network.py:560-574) equal the
    oracle's after every slot - the whole main_test.py slot sequence on the fast path."""
    from oracle.oracle import Oracle, SQ_IEEE
    L = (20.0 if N <= 64 else 10.0) * N + 100
    rd = 2
    cfg = bench_config(N, A, L, reward_design=rd, communication_range=120.0).replace(track_arrival=True)
    rng = np.random.default_rng(5000 + N + A + mode)
    B = 4
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    names = {STEP_MY_STEP: "my_step", STEP_MY_STEP_CH: "my_step_ch", STEP_DESIGN: "my_step_design"}
    env, gen = make_env(cfg, B, mode=names[mode]), make_env(cfg, B, mode=names[mode])
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    for e in (env, gen):
        e.reset_topology(x0, None, v0)
    orc.reset(x0, np.zeros((B, N)), v0)
    for t in range(20):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, _ = env.step(a, t)
        with _general_kernel(gen):
            og, rg, _ = gen.step(a, t)
        o_rew, o_chobs = orc.step(mode, a, t)
        ia, iag, oia = env.info_age(t), gen.info_age(t), orc.info_age(t)
        torch.cuda.synchronize()
        assert torch.equal(obs, og) and torch.equal(rew, rg), t
        assert torch.equal(ia, iag) and np.array_equal(ia.cpu().numpy(), oia), t
    st, sg, oe = env.export_state(), gen.export_state(), orc.export()
    assert torch.equal(st["la"], sg["la"])
    assert np.array_equal(st["la"].cpu().numpy().astype(np.int64), oe["la"])
    env.check()
    gen.check()


def _default_flag_goldens():
    out = []
    for n in golden_names():
        g = Golden(n)
        st, c = g.cfg.State, g.cfg
        if (st.type == 2 and st.add_action and st.action_index == "binary" and st.add_positional_dist_piggy
                and st.add_positional_dist_type == 2 and not (st.add_reward or st.add_index or st.add_velocity
                or st.add_position or st.add_positional_dist or st.add_channel_obs)
                and c.mobility and not c.proportional_fair and not c.enable_fingerprint):
            out.append(n)
    return out


@pytest.mark.parametrize("name", _default_flag_goldens())
def test_reference_fixtures_replayed_on_the_specialised_kernels(name):
    """Every reference fixture recorded with the toy YAML's State flags, replayed through
    the fused `step` (no channel-obs output, so step_fast64 / step_wide run - with the
    arrival stamps, and whichever of my_step / my_step_ch / my_step_design the fixture
    used at each slot) and compared with what the REFERENCE produced: rewards, state
    vectors, positions, information age and the table planes at the recorded checkpoints."""
    g = Golden(name)
    B = 2
    env = make_env(g.cfg, B)
    env.reset_topology(g["x0"], g["y0"], g["v0"])
    ck = g.table_checkpoints()
    cfg = g.cfg
    for i, mode, actions, t, (ep, eps) in g.steps():
        a = env._actions(np.asarray(actions))
        obs, rew, done = env._step(mode, a, t, ep, eps)
        torch.cuda.synchronize()
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for b in range(B):
            if uses_exp(cfg, mode) and mode != STEP_DESIGN:
                assert exp_close(rew[b], g["rews"][i]), (name, i)
            else:
                assert np.array_equal(rew[b], g["rews"][i]), (name, i)
            assert np.array_equal(obs[b], g["state"][i]), (name, i)     # one-hot + histogram: exact
        if i in g.vel_updates:
            env.update_velocity(g.vel_updates[i])
        if g.trace is not None and i == g.trace_after:
            env.load_saved_positions(g.trace)
        ia = env.info_age(t).cpu().numpy()
        st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        for b in range(B):
            assert np.array_equal(st["pos_x"][b], g["pos_x"][i]), (name, i)
            assert np.array_equal(ia[b], g["ia"][i]), (name, i)
            if i in ck:
                j = ck[i]
                assert np.array_equal(st["seq"][b], g["tab_seq"][j])
                assert np.array_equal(st["age"][b], np.minimum(g["tab_age"][j], 255))
                assert np.array_equal(st["x"][b], g["tab_x"][j])
                assert np.array_equal(st["la"][b], g["tab_la"][j])
    env.check()


@pytest.mark.parametrize("N,A", [(64, 32), (256, 64), (128, 64), (40, 7), (200, 33)])
def test_dispatch_guard_specialised_kernels_serve_the_common_configurations(N, A):
    """Dispatch guard (exact: `diral_env_last_kernel`).  step_fast64 / step_wide must serve
    * `step` with the toy YAML's State flags on the PLAIN instantiation,
    * the reference's own call pattern `obs, rews = env.my_step*(a, t)` (channel observation
      requested) and every cheap State flag on a RICH instantiation,
    and the obtain_state that follows my_step* must not launch anything."""
    from diral_amd.config import (KERNEL_CH, KERNEL_EXTRA, KERNEL_FAST64, KERNEL_GENERAL, KERNEL_RICH, KERNEL_RING, KERNEL_WIDE)
    fam = KERNEL_FAST64 if N <= 64 else KERNEL_WIDE
    L = 2000.0 if N <= 64 else 4000.0
    B = 16
    cfg = bench_config(N, A, L)
    env = make_env(cfg, B, dtype=torch.float32)
    env.reset_topology(seed=3)
    a = env.sample(seed=1)
    env.step(a, 0)
    from diral_amd.config import KERNEL_PACKED
    # the xpos ring rides on every specialised launch; the packed table form at N <= 64 and, on dense topologies
    # (N * 2 Rc / L >= 15 neighbours at N <= 128, 20 above), at N > 64
    dense = N * 2 * cfg.communication_range / L >= (15 if N <= 128 else 20)
    ring = KERNEL_RING | (KERNEL_PACKED if (N <= 64 or dense) else 0)
    assert env.last_kernel() == fam | ring                              # plain
    chobs, rew = env.my_step(a, 1)
    assert env.last_kernel() == fam | KERNEL_RICH | ring
    k = env.last_kernel()
    st = env.obtain_state(chobs, a, rew)                                  # served by the fused launch
    assert env.last_kernel() == k and st is env._obs
    env.my_step_ch(a, 2)
    assert env.last_kernel() == fam | KERNEL_RICH | KERNEL_CH | ring
    env.my_step_design(a, 0)
    assert env.last_kernel() == fam | KERNEL_RICH | KERNEL_EXTRA | ring
    env.force_general_kernel(True)
    env.step(a, 3)
    assert env.last_kernel() == KERNEL_GENERAL
    env.check()
    # every cheap State flag at once (the g1_flags_all shape) + State.type 1 + arrival stamps
    flags = dict(add_reward=True, add_index=True, add_velocity=True, add_position=True, add_channel_obs=True)
    for state, extra, want in ((flags, dict(enable_fingerprint=True), fam | KERNEL_RICH),
                               ({}, dict(proportional_fair=True), fam | KERNEL_RICH),
                               (dict(action_index="real", add_channel_obs=True), {}, fam | KERNEL_RICH),
                               (dict(add_action=False), {}, fam | KERNEL_RICH),
                               (flags, dict(track_arrival=True), fam | KERNEL_RICH | KERNEL_EXTRA),
                               # PRR metrics in my_step (DIRAL_F_TRACK_PRR): a run-time switch of the EXTRA instantiations
                               ({}, dict(track_prr=True), fam | KERNEL_EXTRA),
                               # the secondary observation modes: the step on a RICH instantiation, their
                               # columns from posdist_kernel right after it
                               (dict(add_positional_dist=True), {}, fam | KERNEL_RICH),
                               (dict(add_positional_dist_type=1), {}, fam | KERNEL_RICH)):
        c2 = bench_config(N, A, L, State=state, **extra)
        e2 = make_env(c2, B, dtype=torch.float64)
        e2.reset_topology(seed=4)
        e2.step(e2.sample(seed=2), 0)
        assert (e2.last_kernel() & ~ring) == want, (state, extra, e2.last_kernel())
        e2.check()
    # a State block without piggybacked tables (test_env.py:138-139, 231-238: the table-less step), a static
    # topology (network.py:302-305) - run-time switches of the EXTRA instantiations - and a stand-alone
    # obtain_state with foreign arguments (observe_kernel.hpp)
    from diral_amd.config import KERNEL_OBSERVE
    for state, extra, want in ((dict(add_positional_dist_piggy=False), {}, fam | KERNEL_RICH | KERNEL_EXTRA),
                               (dict(add_positional_dist=True, add_positional_dist_piggy=False), {}, fam | KERNEL_RICH | KERNEL_EXTRA),
                               ({}, dict(mobility=False, enable_design_topology=True), fam | KERNEL_EXTRA)):
        c3 = bench_config(N, A, L, State=state, **extra)
        e3 = make_env(c3, 4, dtype=torch.float64)
        e3.reset_topology(seed=5)
        a3 = e3.sample(seed=2)
        e3.step(a3, 0)
        assert (e3.last_kernel() & ~ring) == want, (state, extra, e3.last_kernel())
        e3.obtain_state(None, e3.sample(seed=3), None)
        assert (e3.last_kernel() & 15) == KERNEL_OBSERVE, (state, extra, e3.last_kernel())
        e3.check()
    # what is left for the general kernel: more than 64 resources, vehicles off the common lane at N > 64
    e4 = make_env(bench_config(N, 65, L), 4, dtype=torch.float64)
    e4.reset_topology(seed=5)
    e4.step(e4.sample(seed=2), 0)
    assert e4.last_kernel() == KERNEL_GENERAL
    if N > 64:
        e5 = make_env(bench_config(N, A, L), 4, dtype=torch.float64)
        e5.reset_topology(np.arange(N, dtype=np.float64), np.ones(N), np.full(N, 1.5))
        e5.step(e5.sample(seed=2), 0)
        assert e5.last_kernel() == KERNEL_GENERAL


@pytest.mark.gpu
@pytest.mark.parametrize("N,A,L", [(64, 32, 2000.0), (40, 16, 6000.0), (128, 64, 4000.0), (200, 40, 12000.0)])
def test_realnes_entry_records_round_trip_and_feed_the_step(N, A, L):
    """diral_env_export_entries / diral_env_import_entries: the tables as the RealNeS bridge's
    MA_NeighborTableEntry records (envs/ma_messages_pb2.py:195-230, unpacked at realness_bridge.py:168-191).
    The records must carry exactly export_state's planes (positions narrowed to f32), and an env loaded from
    records must continue bit-for-bit like one loaded from the same values through import_state, on both
    kernel families (the xpos ring is rebuilt from what the records hold)."""
    from diral_amd.vec_env import ENTRY_DTYPE
    assert ENTRY_DTYPE.itemsize == 16
    cfg = bench_config(N, A, L, mobility_vary=True, communication_range=250.0 if L < 5000 else 140.0)
    B = 5
    src, via_rec, via_planes = (make_env(cfg, B, dtype=torch.float64) for _ in range(3))
    src.reset_topology(seed=77)
    rng = np.random.default_rng(N * 3 + A)
    draw = lambda: torch.as_tensor(rng.integers(0, A, size=(B, N)).astype(np.int32), device="cuda:0")
    for t in range(45):
        src.step(draw(), t)
    st = src.export_state()
    rec = src.export_entries()
    host = rec.cpu().numpy().view(ENTRY_DTYPE)[..., 0]
    assert host.shape == (B, N, N)
    assert np.array_equal(host["seq_num"], st["seq"].cpu().numpy())
    assert np.array_equal(host["last_update"], st["age"].cpu().numpy())
    assert np.array_equal(host["pos_x"], st["x"].cpu().numpy().astype(np.float32))
    assert np.array_equal(host["pos_y"], st["y"].cpu().numpy().astype(np.float32))
    assert (host["seq_num"] > 0).any() and (host["seq_num"] == 0).any() == (L > 5000)

    for e in (via_rec, via_planes):
        e.reset_topology(seed=1)
    via_rec.import_state(st["pos_x"], st["pos_y"], st["vel"])
    via_rec.import_entries(host if N != 64 else rec)               # the ndarray and the tensor form
    via_planes.import_state(st["pos_x"], st["pos_y"], st["vel"], seq=st["seq"], age=st["age"],
                            x=st["x"].float().double())
    assert torch.equal(via_rec.export_entries(), rec)              # records survive the round trip
    for t in range(45, 85):
        acts = draw()
        (o1, r1, _), (o2, r2, _) = via_rec.step(acts, t), via_planes.step(acts, t)
        assert torch.equal(o1, o2) and torch.equal(r1, r2), t
        assert via_rec.last_kernel() == via_planes.last_kernel()
    a, b = via_rec.export_state(), via_planes.export_state()
    for k in ("seq", "age", "x", "pos_x"):
        assert torch.equal(a[k], b[k]), k
    with pytest.raises(ValueError):
        via_rec.import_entries(rec[:, :, :-1])


@pytest.mark.gpu
@pytest.mark.parametrize("N,A,L,K,state,vary", [
    (64, 32, 2000.0, 20, dict(add_positional_dist_type=1), False),                   # lane-per-viewer register sort, dense
    (64, 32, 9000.0, 64, dict(add_positional_dist_type=1), True),                    # sparse: few valid entries, stale ones, K = 64,
                                                                                      # equal positions (one speed: exact ties, d = 0)
    (40, 16, 1500.0, 7, dict(add_positional_dist_type=1, add_reward=True), True),    # N < 64: masked rows and lanes
    (2, 2, 100.0, 3, dict(add_positional_dist_type=1, add_positional_dist=True), False),
    (64, 32, 2000.0, 20, dict(add_positional_dist=True), True),                      # one x ranking per env, ties included
    (128, 64, 4000.0, 20, dict(add_positional_dist=True, add_index=True), False),
    (256, 64, 4000.0, 10, dict(add_positional_dist=True, add_action=False, add_positional_dist_piggy=True), False),
    (90, 8, 3000.0, 12, dict(add_positional_dist_type=1, add_positional_dist=True), False),   # N > 64: two lanes per viewer
    (128, 64, 4000.0, 20, dict(add_positional_dist_type=1), True),                   # ... every lane pair full, ties
    (128, 16, 14000.0, 64, dict(add_positional_dist_type=1), False),                 # ... sparse: stale entries, plane + ring
    (200, 40, 5000.0, 9, dict(add_positional_dist_type=1, add_channel_obs=True), False),      # four lanes per viewer, partial
    (256, 64, 4000.0, 20, dict(add_positional_dist_type=1), True),                   # ... full
])
def test_secondary_observation_kernels_vs_oracle(N, A, L, K, state, vary):
    """State.add_positional_dist (sorted signed true distances / norm, network.py:409-430) and
    add_positional_dist_type 1 (weighted np.histogram of the table distances, network.py:432-471) on the
    kernels built for them (csrc/posdist_kernel.hpp): the step on a specialised RICH instantiation, the
    columns from posdist_sorted_flat_kernel / posdist_type1_n64_kernel / posdist_type1_lanes_kernel<4 | 8, 32>, bit for
    bit against the oracle over rollouts with velocity changes; the xpos ring feeds the type-1 kernel."""
    from diral_amd.config import KERNEL_FAST64, KERNEL_WIDE
    cfg = bench_config(N, A, L, mobility_vary=vary, communication_range=250.0 if L < 5000 else 160.0,
                       State=dict(num_bins=K, **state))
    random_rollout(cfg, B=5 if N <= 128 else 3, T=40, seed=700 + N + K, sticky=0.5, vel_every=9, track_prr=False,
                   expect_kernel=KERNEL_FAST64 if N <= 64 else KERNEL_WIDE)


@pytest.mark.parametrize("N,A,L,Rc,form", [(64, 32, 2000.0, 250.0, None), (40, 9, 1500.0, 250.0, None), (64, 16, 9000.0, 140.0, None),
                                           (33, 5, 6000.0, 100.0, None), (128, 64, 4000.0, 250.0, None), (96, 12, 20000.0, 120.0, None),
                                           (128, 64, 4000.0, 250.0, "plane"), (96, 12, 20000.0, 120.0, "packed"), (100, 20, 2500.0, 250.0, None),
                                           (100, 20, 2500.0, 250.0, "plane"),
                                           (256, 64, 4000.0, 250.0, None), (256, 64, 4000.0, 250.0, "plane"),
                                           (200, 24, 30000.0, 140.0, None), (200, 24, 30000.0, 140.0, "packed"),
                                           (256, 16, 12000.0, 250.0, "packed")])
def test_packed_tables_and_xpos_ring_agree_with_the_oracle_through_every_consumer(N, A, L, Rc, form, monkeypatch):
    """The specialised kernels do not keep the reference-shaped planes current: N <= 64 stores the table as
    thermometer codes + ages + own sequence numbers, and both families keep the xpos of young entries in the
    per-subject ring (csrc/step_fast64.hpp, step_wide.hpp); the planes `tkey` / `tx` only answer for entries older
    than 7 stamps.  Everything else - export_state, a stand-alone obtain_state with foreign arguments
    (diral_env_observe), a step of the general kernel, import_state - must see the planes completed first, and the
    packed state must be rebuilt when another kernel moved the tables.  The comparator is the oracle: a rollout with
    all of those mixed in must match it slot by slot (state, reward, channel observation) and plane by plane at
    every export.  The sparse topologies (Rc < 200) hold entries beyond the codes / the ring (keyed quads,
    byte-rank / 32-bit passes, hand-over at lag 7)."""
    from oracle.oracle import Oracle, SQ_IEEE
    from diral_amd.config import KERNEL_FAST64, KERNEL_GENERAL, KERNEL_PACKED, KERNEL_RING, KERNEL_WIDE
    cfg = bench_config(N, A, L, mobility_vary=True, communication_range=Rc)
    # the table form of a 64 < N <= 256 handle follows the density (N * 2 Rc / L >= 20 neighbours, 15 at N <= 128: packed);
    # `form` forces the other one (DIRAL_TABLE_FORM, read at create): both forms on both kinds of topology
    if form:
        monkeypatch.setenv("DIRAL_TABLE_FORM", form)
    want_packed = N <= 64 or form == "packed" or (form is None and N * 2 * Rc / L >= (15 if N <= 128 else 20))
    B = 4
    rng = np.random.default_rng(N * 7 + A)
    x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
    v0 = np.full((B, N), 1.7)
    env = make_env(cfg, B)
    env.reset_topology(x0, 0.0, v0)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    orc.reset(x0, np.zeros((B, N)), v0)

    def same_tables(t):
        st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        oe = orc.export()
        for k in ("seq", "x", "pos_x", "vel"):
            assert np.array_equal(st[k], oe[k]), (k, t)
        assert np.array_equal(st["age"], np.minimum(oe["age"], 255)), t

    max_lag = 0
    for t in range(75):
        acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
        general = t in (13, 14, 37) or 55 <= t < 58                  # the general kernel takes over for some slots
        env.force_general_kernel(general)
        obs, rew, chobs, _ = gpu_step(env, STEP_MY_STEP, acts, t)
        assert (env.last_kernel() & (15 | KERNEL_RING)) == (KERNEL_GENERAL if general else _fam(N) | KERNEL_RING)
        assert general or bool(env.last_kernel() & KERNEL_PACKED) == want_packed, env.last_kernel()
        o_rew, o_chobs = orc.step(STEP_MY_STEP, acts, t)
        o_state = orc.obtain_state(acts, o_chobs, o_rew)
        assert np.array_equal(obs, o_state), (t, np.argwhere(obs != o_state)[:5])
        assert np.array_equal(rew, o_rew) and np.array_equal(chobs, o_chobs), t
        if t % 7 == 3:
            same_tables(t)                                           # export: the plane completed from the ring
        if t % 9 == 5:                                               # foreign arguments: a diral_env_observe launch
            other = rng.integers(0, A, size=(B, N)).astype(np.int32)
            fake = np.full((B, N), 0.25)
            s1 = env.obtain_state(None, other, fake).cpu().numpy()
            s2 = orc.obtain_state(other, np.zeros((B, N, A)), fake)
            assert np.array_equal(s1, s2), t
        if t in (26, 60):                                            # checkpoint round trip through import_state
            st = env.export_state()
            lag = torch.diagonal(st["seq"], dim1=1, dim2=2).unsqueeze(1) - st["seq"]
            max_lag = max(max_lag, int(lag[st["seq"] > 0].max().item()))
            env.import_state(st["pos_x"], st["pos_y"], st["vel"], seq=st["seq"], age=st["age"], x=st["x"])
        if t % 25 == 24:
            draws = rng.integers(1, 4, size=(B, N)).astype(np.uint8)
            env.update_velocity(draws)
            orc.update_velocity(draws)
    same_tables(75)
    if Rc < 200:
        assert max_lag > 7, max_lag                                  # the sparse topologies do leave the ring
    env.check()


def test_import_of_conflicting_tables_is_reported():
    """diral_env_import_state / import_entries accept any tables; the ring-based kernels need entries about one
    subject with equal sequence numbers to carry equal xpos (no run of the reference violates it).  A violation is
    detected at import and surfaces in diral_env_check as DIRAL_ERR_TABLE_CONFLICT."""
    from diral_amd.config import ERR_TABLE_CONFLICT
    from diral_amd.vec_env import DiralError
    for N, A in ((64, 32), (128, 64)):
        cfg = bench_config(N, A, 3000.0)
        env = make_env(cfg, 3)
        env.reset_topology(seed=9)
        for t in range(12):
            env.step(env.sample(seed=t), t)
        st = env.export_state()
        env.import_state(st["pos_x"], st["pos_y"], st["vel"], seq=st["seq"], age=st["age"], x=st["x"])
        env.check()                                                  # a reachable state imports cleanly
        seq, x = st["seq"].clone(), st["x"].clone()
        # two viewers of env 1 hold subject 5 at the SAME sequence number ...
        own = int(seq[1, 5, 5].item())
        seq[1, 2, 5] = own - 1
        seq[1, 3, 5] = own - 1
        x[1, 2, 5] = 111.0
        x[1, 3, 5] = 222.0                                           # ... with different xpos
        env.import_state(st["pos_x"], st["pos_y"], st["vel"], seq=seq, age=st["age"], x=x)
        with pytest.raises(DiralError) as ei:
            env.check()
        assert ei.value.status == ERR_TABLE_CONFLICT
        env.import_state(st["pos_x"], st["pos_y"], st["vel"], seq=st["seq"], age=st["age"], x=st["x"])
        env.check()


def test_kernel_path_switch_inside_a_graph_capture_is_an_error():
    """A ring <-> plane conversion is a launch that depends on host-side validity flags; recorded into a hipGraph it
    would replay against tables it no longer describes.  The call fails with DIRAL_ERR_CAPTURE instead (and the
    capture of plain steps keeps working: test_step_is_capturable_in_a_hip_graph)."""
    from diral_amd.config import ERR_CAPTURE
    from diral_amd.vec_env import DiralError
    cfg = c2_config()
    env = make_env(cfg, 4, dtype=torch.float32)
    env.reset_topology(seed=2)
    acts = env.sample(seed=1)
    for t in range(3):
        env.step(acts, t)                                            # ring steps: the plane is now incomplete
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    env.force_general_kernel(True)                                   # the next step needs the plane materialised
    with torch.cuda.stream(s):
        g.capture_begin()
        try:
            keep = acts + 0                                          # (the captured graph is not empty)
            with pytest.raises(DiralError) as ei:
                env.step(acts, 3)
            assert ei.value.status == ERR_CAPTURE
        finally:
            g.capture_end()
    torch.cuda.synchronize()
    env.force_general_kernel(False)
    env.step(acts, 3)                                                # nothing was launched or recorded: the env goes on
    env.check()


@pytest.mark.parametrize("N,A,y", [(64, 32, False), (40, 7, True), (128, 64, False), (256, 16, False), (200, 33, True)])
def test_observe_kernel_f32_is_the_cast_of_f64_and_leaves_the_env_alone(N, A, y):
    """diral_env_observe (observe_kernel.hpp): float32 output == float32(float64 output); the call changes nothing
    in the env (tables, positions, metrics); vehicles off the common lane take the non-FLAT instantiation."""
    from diral_amd.config import KERNEL_OBSERVE
    cfg = bench_config(N, A, 25.0 * N + 100, State=dict(add_channel_obs=True, add_reward=True, add_position=True,
                                                        add_velocity=True, add_index=True), enable_fingerprint=True)
    rng = np.random.default_rng(N + A)
    B = 5
    x0 = rng.integers(0, int(cfg.highway_length), size=(B, N)).astype(np.float64)
    y0 = rng.integers(0, 2, size=(B, N)).astype(np.float64) if y else np.zeros((B, N))
    v0 = rng.uniform(1.1, 2.7, size=(B, N))
    e64, e32 = make_env(cfg, B, dtype=torch.float64), make_env(cfg, B, dtype=torch.float32)
    for e in (e64, e32):
        e.reset_topology(x0, y0, v0)
    for t in range(14):
        a = rng.integers(0, A, size=(B, N)).astype(np.int32)
        e64.step(a, t)
        e32.step(a, t)
    before = {k: v.clone() for k, v in e64.export_state().items()}
    m0 = e64.metrics().clone()
    fa = rng.integers(0, A, size=(B, N)).astype(np.int32)
    fc = rng.uniform(0.0, 300.0, size=(B, N, A))
    fr = rng.uniform(-3.0, 1.0, size=(B, N))
    s64 = e64.obtain_state(fc, fa, fr, 7.0, 0.5).clone()
    s32 = e32.obtain_state(fc, fa, fr, 7.0, 0.5).clone()
    assert (e64.last_kernel() & 15) == KERNEL_OBSERVE and (e32.last_kernel() & 15) == KERNEL_OBSERVE
    assert torch.equal(s64.to(torch.float32), s32)
    after = e64.export_state()
    for k in before:
        assert torch.equal(before[k], after[k]), k
    assert torch.equal(m0, e64.metrics())
    e64.check(); e32.check()


@pytest.mark.parametrize("N,A,G,B", [(64, 32, 4, 64), (64, 32, 3, 50), (128, 64, 2, 12), (256, 64, 2, 6)])
def test_streamed_sub_batches_equal_one_handle(N, A, G, B):
    """StreamedVecEnv (diral_amd/streamed.py): the batch stepped as G sub-batches on G streams - with and without a
    per-slot wait on the caller's stream - against ONE handle holding all B envs: same device-drawn topology (global
    env index), same states, rewards, channel observations, tables and metrics, bit for bit."""
    from diral_amd.streamed import StreamedVecEnv
    cfg = bench_config(N, A, 30.0 * N + 100, mobility_vary=True)
    one = make_env(cfg, B, dtype=torch.float32)
    one.reset_topology(seed=77)
    for sync in (True, False):
        one.reset_topology(seed=77)
        many = StreamedVecEnv(cfg, batch=B, groups=G, out_dtype=torch.float32)
        many.reset_topology(seed=77)
        acts = [one.sample(seed=100 + i) for i in range(30)]
        outs = []
        for t in range(30):
            o1, r1, d1 = one._step(STEP_MY_STEP, acts[t], t, want_chobs=True)
            o2, r2, d2 = many.step(acts[t], t, sync=sync)
            if sync:
                assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(one._chobs, many.chobs), t
            if t % 10 == 9:
                one.update_velocity(seed=t)
                many.update_velocity(seed=t)
        many.wait()
        assert torch.equal(o1, many.obs) and torch.equal(r1, many.rew) and torch.equal(one._chobs, many.chobs)
        a, b = one.export_state(), many.export_state()
        for k in a:
            assert torch.equal(a[k], b[k]), k
        assert torch.equal(one.metrics(clear=True)[:, [0, 2, 3]], many.metrics()[:, [0, 2, 3]])
        many.check()
        many.close()
    one.check()


@pytest.mark.parametrize("K", [10, 12])
@pytest.mark.parametrize("N,A,B,ch", [(64, 32, 64, False), (64, 32, 16, True), (128, 64, 8, False), (256, 64, 4, False)])
def test_graph_rollout_equals_eager(N, A, B, ch, K):
    """diral_amd/rollout.py: K slots of [env step -> reward shaping -> SPS policy] captured into one hipGraph and replayed
    20 times against the same 21 K slots run eagerly: slot number (done flag, arrival stamps), policy draws and actions come
    from device memory, so the replays move on exactly like the eager loop - env state, tables, policy state, metrics and the
    last outputs equal bit for bit.  K = 12 (a multiple of 3): the captured launches of step_fast64 rotate the slow-env
    sets (diral_env_set_capture_rotation) instead of reading a frozen list - every env still runs exactly once per slot."""
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    cfg = bench_config(N, A, 30.0 * N + 100, reward_design=2, track_arrival=ch)
    runs = []
    for capture in (True, False):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32, io_ring=2)
        env.reset_topology(seed=5)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=3)
        ro = GraphRollout(env, pol, K=K, enable_channel=ch, capture=capture)
        assert (ro._phase is not None) == (capture and K % 3 == 0)
        ro.run(20 if capture else 21)
        torch.cuda.synchronize()
        assert ro.slots == 21 * K and ro.clock.value() == 21 * K
        runs.append((env, pol, ro))
    (e1, p1, r1), (e2, p2, r2) = runs
    a, b = e1.export_state(), e2.export_state()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.equal(p1.counter, p2.counter)
    for x, y in zip(r1.last(), r2.last()):
        assert torch.equal(x, y)
    assert torch.equal(e1._done, e2._done) and torch.equal(e1._chobs, e2._chobs)
    m1, m2 = e1.metrics(), e2.metrics()
    assert torch.equal(m1[:, [0, 2, 3]], m2[:, [0, 2, 3]]) and torch.allclose(m1, m2, rtol=1e-12, atol=1e-9)
    assert float(m1[:, 0].min()) == 21.0 * K
    if ch:
        assert torch.equal(e1.info_age(21 * K - 1), e2.info_age(21 * K - 1))
    for e, _, r in runs:
        e.check()
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,A,B,dt,fused_expected", [
    (64, 32, 96, torch.float32, True),       # the metric's shape: ONE launch per slot
    (64, 32, 33, torch.float64, True),
    (40, 20, 50, torch.float32, True),       # padded lanes, A not a multiple of 32
    (9, 5, 40, torch.float32, True),         # np.sum's eight accumulators + a sequential tail of one
    (6, 4, 40, torch.float32, False),        # N < 8: np.sum is sequential - the three launches
    (96, 48, 12, torch.float32, False),      # N > 64: step_wide + the two policy launches
])
def test_fused_policy_slot_equals_three_launches(N, A, B, dt, fused_expected):
    """diral_env_step_policy (env step + reward shaping + SPS decision as ONE launch, the channel observation handed
    over in LDS) against the same closed loop as three launches (diral_env_step with the channel observation,
    diral_driver_shape, diral_sps_step_chobs_clocked): 120 slots each, every per-slot output compared bit for bit -
    state, raw and shaped rewards, sum / collision columns, done, the next actions - and at the end the env (tables,
    positions), the policy state and the metrics."""
    from diral_amd.config import KERNEL_POLICY
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    cfg = bench_config(N, A, 30.0 * N + 100, reward_design=2)
    runs = []
    for fused in (True, False):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=dt, io_ring=2)
        env.reset_topology(seed=11)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=4)
        ro = GraphRollout(env, pol, K=2, capture=False, fused=fused)
        runs.append((env, pol, ro))
    (e1, p1, r1), (e2, p2, r2) = runs
    for step in range(60):
        r1.run(1)
        r2.run(1)
        assert bool(e1.last_kernel() & KERNEL_POLICY) == fused_expected
        assert not (e2.last_kernel() & KERNEL_POLICY)
        for x, y in zip(r1.last(), r2.last()):
            assert torch.equal(x, y), step
        assert torch.equal(e1._rew, e2._rew) and torch.equal(e1._done, e2._done), step
        for i in (0, 1):
            assert torch.equal(r1.shaped[i], r2.shaped[i]) and torch.equal(r1.sum_r[i], r2.sum_r[i]), step
            assert torch.equal(r1.coll[i], r2.coll[i]) and torch.equal(r1.actions[i], r2.actions[i]), step
    a, b = e1.export_state(), e2.export_state()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.equal(p1.counter, p2.counter)
    m1, m2 = e1.metrics(), e2.metrics()
    assert torch.equal(m1, m2)
    for e, _, r in runs:
        e.check()
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw,ch", [(dict(track_arrival=True), False), (dict(), True), (dict(track_prr=True), False)])
def test_policy_slot_falls_back_to_three_launches_outside_the_fused_instantiation(kw, ch):
    """`step_policy` on configurations the POL instantiation does not take (arrival stamps, my_step_ch, PRR metrics): the
    library says so before it launches anything, the call is repeated with a channel-observation buffer, and the slot
    equals the explicit three launches."""
    from diral_amd.config import KERNEL_POLICY
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    cfg = bench_config(64, 32, 2000.0, reward_design=2, **kw)
    B, N, A = 24, 64, 32
    runs = []
    for fused in (True, False):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", io_ring=2)
        env.reset_topology(seed=8)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=6)
        runs.append((env, pol, GraphRollout(env, pol, K=2, capture=False, fused=fused, enable_channel=ch)))
    (e1, p1, r1), (e2, p2, r2) = runs
    for step in range(25):
        r1.run(1)
        r2.run(1)
        assert not (e1.last_kernel() & KERNEL_POLICY)
        for x, y in zip(r1.last(), r2.last()):
            assert torch.equal(x, y), step
        assert torch.equal(e1._rew, e2._rew) and torch.equal(e1._chobs, e2._chobs)
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.equal(p1.counter, p2.counter)
    a, b = e1.export_state(), e2.export_state()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for e, _, r in runs:
        e.check()
        r.close()


def test_fused_policy_slot_with_the_observation_written_and_captured():
    """The fused slot also writes the channel observation when asked to (equal to the three-launch one), and a K-slot
    hipGraph of fused slots replays like the eager loop."""
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    cfg = c2_config()
    B, N, A = 64, cfg.num_users, cfg.num_channels
    # (a) chobs_out given
    e1 = VecV2VEnv(cfg, batch=B, device="cuda:0")
    e2 = VecV2VEnv(cfg, batch=B, device="cuda:0")
    p1, p2 = SpsPolicy(B, N, A, seed=9), SpsPolicy(B, N, A, seed=9)
    for e in (e1, e2):
        e.reset_topology(seed=2)
    a1, a2 = p1.prev_action.clone(), p2.prev_action.clone()
    n1, n2 = torch.empty_like(a1), torch.empty_like(a2)
    sh1 = torch.empty((B, N), dtype=torch.float32, device="cuda:0")
    for t in range(30):
        e1.step_policy(a1, t, p1, n1, shaped_out=sh1, want_chobs=True)
        e2._step(0, a2, t, want_chobs=True)
        p2.step_from_chobs(e2._chobs, a2, out=n2)
        assert torch.equal(e1._chobs, e2._chobs) and torch.equal(n1, n2) and torch.equal(e1._obs, e2._obs), t
        a1, n1 = n1, a1
        a2, n2 = n2, a2
    # (b) captured
    runs = []
    for capture in (True, False):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", io_ring=2)
        env.reset_topology(seed=5)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=3)
        ro = GraphRollout(env, pol, K=10, capture=capture, fused=True)
        ro.run(6 if capture else 7)
        torch.cuda.synchronize()
        runs.append((env, pol, ro))
    (g1, q1, r1), (g2, q2, r2) = runs
    sa, sb = g1.export_state(), g2.export_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(q1.prev_action, q2.prev_action) and torch.equal(q1.counter, q2.counter)
    for x, y in zip(r1.last(), r2.last()):
        assert torch.equal(x, y)
    for e, _, r in runs:
        e.check()
        r.close()


@pytest.mark.parametrize("N,A,K", [(256, 64, 48), (200, 48, 40)])
def test_stand_alone_obtain_state_with_more_than_64_kb_of_lds(N, A, K):
    """observe_kernel's dynamic LDS grows with N x K (ring rows + histogram rows): ~66 KB at N = 256 / K = 48 (more bins do not fit the general kernel at that size),
    ~66 KB at N = 200 / K = 40 - past the 64 KB a kernel gets without hipFuncAttributeMaxDynamicSharedMemorySize
    (set per handle at create).  `random_rollout` calls obtain_state with foreign arguments every six slots."""
    random_rollout(bench_config(N, A, 16.0 * N, State=dict(num_bins=K)), B=3, T=14, seed=300 + N)


@pytest.mark.parametrize("N,A", [(64, 32), (48, 6), (128, 64), (256, 64)])
def test_closest_transmitter_on_ties_and_at_the_range_boundary(N, A):
    """Network.find_closest_tx (network.py:378-398) on the plain my_step instantiations - no arrival stamps, no PRR
    tracking: step_fast64's search keeps a running minimum over ALL transmitters and tests the range once per resource
    (csrc/step_fast64.hpp, P1).  A standing topology on a 50 m grid with Rc = 250 m makes the cases it must not get
    wrong permanent: transmitters exactly Rc away (out of range: strict '<'), two transmitters at the same distance on
    either side or at the same spot (the lower id wins), a vehicle on top of its transmitter (distance 0).  State, reward
    and the channel observation (the distance to the chosen transmitter, 100000 when none is in range) against the
    oracle, bit for bit, and the tables at the end."""
    from oracle.oracle import Oracle, SQ_IEEE
    cfg = bench_config(N, A, 50.0 * 40, communication_range=250.0)
    rng = np.random.default_rng(77 + N)
    B = 12
    x0 = (50.0 * rng.integers(0, 40, size=(B, N))).astype(np.float64)
    y0 = np.zeros((B, N))
    v0 = np.zeros((B, N))                                  # nobody moves: the grid stays a grid
    env = make_env(cfg, B)
    env.reset_topology(x0, y0, v0)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=8)
    orc.reset(x0, y0, v0)
    seen_boundary = seen_tie = False
    for t in range(12):
        acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
        obs, rew, chobs, _ = gpu_step(env, STEP_MY_STEP, acts, t)
        assert (env.last_kernel() & 15) == _fam(N)
        o_rew, o_chobs = orc.step(STEP_MY_STEP, acts, t)
        o_state = orc.obtain_state(acts, o_chobs, o_rew)
        assert np.array_equal(rew, o_rew), t
        assert np.array_equal(chobs, o_chobs), (t, np.argwhere(chobs != o_chobs)[:5])
        assert np.array_equal(obs, o_state), (t, np.argwhere(obs != o_state)[:5])
        # the slot really held the cases: a (viewer, transmitter) pair exactly Rc apart, and a viewer with two
        # equidistant nearest transmitters on one resource
        for b in range(2):
            for i in range(A):
                tx = np.flatnonzero(acts[b] == i)
                if len(tx) == 0:
                    continue
                d = np.abs(x0[b][:, None] - x0[b][tx][None, :])
                seen_boundary |= bool(np.any(d == 250.0))
                if len(tx) > 1:
                    ds = np.sort(d, axis=1)
                    seen_tie |= bool(np.any((ds[:, 0] == ds[:, 1]) & (ds[:, 0] < 250.0)))
    assert seen_boundary and seen_tie
    st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
    oe = orc.export()
    for k in ("pos_x", "seq", "x"):
        assert np.array_equal(st[k], oe[k]), k
    assert np.array_equal(st["age"], np.minimum(oe["age"], 255))
    env.check()


def test_graph_replays_with_rotating_slow_sets_survive_eager_steps_in_between():
    """A captured rollout whose step launches rotate the slow-env sets (K = 6: diral_env_set_capture_rotation), with
    plain eager steps of the same env between the replays - one, then two, so that the launch phase is off by one and by
    two when the next replay starts: `GraphRollout.run` realigns it (diral_env_align_phase empties the sets), and every
    env still runs exactly once in every launch.  Against the same sequence without a graph, bit for bit.  Sticky
    actions make a third of the envs slow, so the lists are long and change from slot to slot."""
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    N, A, B, K = 64, 32, 256, 6
    cfg = bench_config(N, A, 30.0 * N + 100, reward_design=2)
    runs = []
    for capture in (True, False):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32, io_ring=2)
        env.reset_topology(seed=9)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=4)
        pol.keep_prob = 0.95
        ro = GraphRollout(env, pol, K=K, capture=capture)
        assert (ro._phase is not None) == capture
        extra = [env.sample(seed=50 + i) for i in range(3)]
        ro.run(15 if capture else 16)                                    # (the capture ran its K slots once eagerly)
        env._step(ro.mode, extra[0], 0, want_chobs=True)                 # one eager launch: phase + 1
        ro.run(5)
        env._step(ro.mode, extra[1], 1, want_chobs=True)                 # two: phase + 2
        env._step(ro.mode, extra[2], 2, want_chobs=True)
        ro.run(5)
        env._step(ro.mode, extra[0], 3, want_chobs=True)                 # a last eager step: its outputs are compared
        torch.cuda.synchronize()
        runs.append((env, pol, ro))
    (e1, p1, r1), (e2, p2, r2) = runs
    a, b = e1.export_state(), e2.export_state()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.equal(p1.counter, p2.counter)
    assert torch.equal(e1._obs, e2._obs) and torch.equal(e1._rew, e2._rew) and torch.equal(e1._chobs, e2._chobs)
    m1, m2 = e1.metrics(), e2.metrics()
    assert torch.equal(m1[:, [0, 2, 3]], m2[:, [0, 2, 3]])
    assert float(m1[:, 0].min()) == float(m1[:, 0].max()) == 26.0 * K + 4          # (the eager pass of the capture, 25 runs, 4 steps)
    for e, _, r in runs:
        e.check()
        r.close()


@pytest.mark.parametrize("K,N,A,L", [(10, 64, 32, 2020.0), (4, 64, 32, 2020.0), (4, 128, 64, 4000.0)])
def test_graph_replays_on_a_frozen_slow_set_survive_eager_steps_in_between(K, N, A, L):
    """A captured rollout whose launch count is NOT a multiple of three cannot rotate the slow-env sets: every captured
    launch is baked with the set the last eager launch before the capture read.  Eager steps between two replays keep
    rotating through that set - one of them clears it, the next rebuilds it.  A set must be either a complete list or
    EMPTY at every launch boundary (count and flags: step_fast64.hpp), else a replay issued two eager steps after the
    capture finds a zero count under standing flags and steps the flagged envs not at all.  One, two
    and three eager steps between replays; against the same sequence without a graph, bit for bit; every env's slot
    counter advanced by every launch.  Sticky actions keep a third of the envs on the list.  The 128-vehicle case runs
    step_wide's packed form, whose envs with a broken highway (one in ten at this density) ask for the first blocks the same
    way (csrc/step_wide.hpp)."""
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    B = 256
    cfg = bench_config(N, A, L, reward_design=2)
    runs = []
    for capture in (True, False):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32, io_ring=2)
        env.reset_topology(seed=9)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=4)
        pol.keep_prob = 0.95
        # (eager warm-up first: the tables age, the lists fill - the capture then bakes a set with standing flags)
        warm = [env.sample(seed=70 + i) for i in range(2)]
        for i in range(40):
            env._step(0, warm[i & 1], i, want_chobs=True)
        ro = GraphRollout(env, pol, K=K, capture=capture)
        assert ro._phase is None                                         # no rotation inside this graph
        extra = [env.sample(seed=50 + i) for i in range(3)]
        slots = 40 + K
        ro.run(3 if capture else 4)                                      # (the capture ran its K slots once eagerly)
        slots += 3 * K
        for n_eager in (1, 2, 3, 2, 1):
            for j in range(n_eager):
                env._step(ro.mode, extra[j], j, want_chobs=True)
            ro.run(2)
            slots += n_eager + 2 * K
            m = env.metrics()
            assert float(m[:, 0].min()) == float(m[:, 0].max()) == float(slots), (capture, n_eager, m[:, 0].min(), slots)
        env._step(ro.mode, extra[0], 3, want_chobs=True)                 # a last eager step: its outputs are compared
        torch.cuda.synchronize()
        runs.append((env, pol, ro))
    (e1, p1, r1), (e2, p2, r2) = runs
    a, b = e1.export_state(), e2.export_state()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # (the lists were not empty: some envs hold entries 7 or more stamps behind their subject - the ones that ask for a
    # place among the first blocks - and some hold none)
    seq = a["seq"]
    lagged = ((seq > 0) & (torch.diagonal(seq, dim1=1, dim2=2).unsqueeze(1) - seq >= 7)).flatten(1).any(dim=1)
    assert 0 < int(lagged.sum()) < B, int(lagged.sum())
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.equal(p1.counter, p2.counter)
    assert torch.equal(e1._obs, e2._obs) and torch.equal(e1._rew, e2._rew) and torch.equal(e1._chobs, e2._chobs)
    m1, m2 = e1.metrics(), e2.metrics()
    assert torch.equal(m1[:, [0, 2, 3]], m2[:, [0, 2, 3]])
    for e, _, r in runs:
        e.check()
        r.close()


def test_step_policy_rejects_tensors_the_c_abi_would_misread():
    """`VecV2VEnv.step_policy` hands raw pointers to diral_env_step_policy: dtype, shape, contiguity, device and
    aliasing of the action tensors are checked in Python."""
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    cfg = c2_config()
    B, N, A = 8, cfg.num_users, cfg.num_channels
    env = VecV2VEnv(cfg, batch=B, device="cuda:0")
    env.reset_topology(seed=1)
    pol = SpsPolicy(B, N, A, seed=2)
    a = pol.prev_action.clone()
    out = torch.empty_like(a)
    with pytest.raises(ValueError):
        env.step_policy(a.long(), 0, pol, out)
    with pytest.raises(ValueError):
        env.step_policy(a, 0, pol, a)
    with pytest.raises(ValueError):
        env.step_policy(a.t().contiguous().t(), 0, pol, out)
    with pytest.raises(ValueError):
        env.step_policy(a, 0, pol, out[:, : N - 1])
    with pytest.raises(ValueError):
        env.step_policy(a, 0, pol, out, shaped_out=torch.empty((B, N), dtype=torch.float64, device="cuda:0"))
    env.step_policy(a, 0, pol, out)
    env.check()


@pytest.mark.parametrize("vary,K,t0,want_obs", [(False, 5, 0, True), (False, 25, 3, False), (True, 25, 10, True), (True, 5, 22, False),
                                                (True, 6, 19, True)])
def test_k_slots_in_one_launch_equal_k_one_slot_launches(vary, K, t0, want_obs):
    """`diral_env_step_policy` with DiralSlotPolicy::slots = K (step_fast64_slots_kernel: the env stays in registers and
    LDS from slot to slot, no table traffic, no histogram unless the last slot's state is asked for) against K fused
    one-slot launches - and, with mobility_vary, `update_velocity(seed = vel_seed + episode)` behind every slot that ends
    an episode (main_test.py:226-233): per-slot shaped rewards / sums / collisions, the last slot's state, reward, done,
    the next actions, the policy state, the metrics and the exported tables, positions and velocities, bit for bit.  A
    warm-up puts entries beyond the codes (keyed quads) into the tables first; two K-slot launches back to back, then a
    one-slot launch on both sides."""
    from diral_amd.config import KERNEL_POLICY
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    N, A, B = 64, 32, 96
    cfg = bench_config(N, A, 30.0 * N + 100, reward_design=2, mobility_vary=vary)
    vel_seed = 777
    runs = []
    for fused_k in (False, True):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32)
        env.reset_topology(seed=11)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=5)
        pol.keep_prob = 0.9
        a = pol.prev_action.clone()
        nxt = torch.empty_like(a)
        t = 0
        for _ in range(t0):                                   # warm-up, one slot per launch on both sides
            env.step_policy(a, t, pol, nxt)
            if vary and t % cfg.episode_interval == cfg.episode_interval - 1:
                env.update_velocity(seed=vel_seed + t // cfg.episode_interval)
            a, nxt = nxt, a
            t += 1
        outs = []
        for rep in range(2):
            sh = torch.zeros((K, B, N), dtype=torch.float32, device="cuda:0")
            sr = torch.zeros((K, B), dtype=torch.float32, device="cuda:0")
            co = torch.zeros((K, B), dtype=torch.float32, device="cuda:0")
            if fused_k:
                env.step_policy(a, t, pol, nxt, shaped_out=sh, sum_r_out=sr, collision_out=co, slots=K, vel_seed=vel_seed,
                                want_obs=want_obs)
                assert env.last_kernel() & KERNEL_POLICY
                a, nxt = nxt, a
                t += K
            else:
                for k in range(K):
                    env.step_policy(a, t, pol, nxt, shaped_out=sh[k], sum_r_out=sr[k], collision_out=co[k])
                    if vary and t % cfg.episode_interval == cfg.episode_interval - 1:
                        env.update_velocity(seed=vel_seed + t // cfg.episode_interval)
                    a, nxt = nxt, a
                    t += 1
            outs.append((sh, sr, co, env._obs.clone(), env._rew.clone(), env._done.clone(), a.clone()))
        env.step_policy(a, t, pol, nxt)                        # a one-slot launch behind the K-slot ones
        torch.cuda.synchronize()
        runs.append((env, pol, outs, nxt.clone()))
    (e1, p1, o1, n1), (e2, p2, o2, n2) = runs
    for rep in range(2):
        for i, name in enumerate(("shaped", "sum_r", "collisions")):
            assert torch.equal(o1[rep][i], o2[rep][i]), (rep, name)
        if want_obs:
            assert torch.equal(o1[rep][3], o2[rep][3]), rep
        assert torch.equal(o1[rep][4], o2[rep][4]) and torch.equal(o1[rep][5], o2[rep][5]) and torch.equal(o1[rep][6], o2[rep][6]), rep
    assert torch.equal(n1, n2) and torch.equal(e1._obs, e2._obs) and torch.equal(e1._rew, e2._rew)
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.equal(p1.counter, p2.counter)
    sa, sb = e1.export_state(), e2.export_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(e1.metrics(), e2.metrics())
    if vary:
        assert not torch.equal(sa["vel"], torch.full_like(sa["vel"], 1.7))      # an episode ended inside the launches
    e1.check()
    e2.check()


@pytest.mark.parametrize("case", ["rich_vel", "rich_vel_last_ends", "f64_chobs", "stuck_penalty", "prop_fair", "slot_clock",
                                  "sorted_distances"])
def test_k_slots_in_one_launch_on_the_other_code_paths_of_the_slot_loop(case):
    """The K-slot kernel's separate code beside the plain float32 state: the RICH output tail (add_position / add_reward /
    add_index / add_velocity - with mobility_vary the velocity column must show the velocities update_velocity left
    INSIDE the launch, also when the last slot itself ends an episode -, the channel observation written out), float64
    outputs, the stuck-action penalty (pen_counter / pen_prev_actions read-modify-written every slot), the
    proportional-fair counters, the device slot clock, and the secondary observation launch behind a K-slot step - each
    against K one-slot calls, bit for bit."""
    from diral_amd.config import KERNEL_POLICY
    from diral_amd.rollout import SlotClock
    from diral_amd.sps import SpsPolicy
    from diral_amd.vec_env import VecV2VEnv
    N, A, B = 64, 32, 48
    kw, dt, K, t0, vary = {}, torch.float32, 6, 21, False
    if case in ("rich_vel", "rich_vel_last_ends"):
        vary = True
        kw = dict(State=dict(add_velocity=True, add_reward=True, add_position=True, add_index=True, add_channel_obs=True))
        K, t0 = (6, 21) if case == "rich_vel" else (4, 21)        # episode ends at t = 24: inside the launch / its last slot
    elif case == "f64_chobs":
        dt, kw = torch.float64, dict(State=dict(add_channel_obs=True))
    elif case == "prop_fair":
        kw = dict(proportional_fair=True)
    elif case == "sorted_distances":
        kw = dict(State=dict(add_positional_dist=True))
    cfg = bench_config(N, A, 30.0 * N + 100, reward_design=2, mobility_vary=vary, **kw)
    vel_seed = 4242
    want_chobs = case in ("f64_chobs", "rich_vel")
    runs = []
    for fused_k in (False, True):
        env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=dt)
        env.reset_topology(seed=13)
        pol = SpsPolicy(B, N, A, device="cuda:0", seed=9)
        pol.keep_prob = 0.95 if case in ("stuck_penalty", "prop_fair") else 0.8
        clock = SlotClock("cuda:0", 0) if case == "slot_clock" else None
        if clock is not None:
            env.set_clock(clock.t)
        pen = None
        if case == "stuck_penalty":
            pen = (2, -10.0, torch.zeros((B, N), dtype=torch.int32, device="cuda:0"),
                   torch.full((B, N), -1, dtype=torch.int32, device="cuda:0"))
        a = pol.prev_action.clone()
        nxt = torch.empty_like(a)
        t = 0

        def one(sh=None, sr=None, co=None):
            nonlocal a, nxt, t
            # (with a device clock the by-value slot number is an offset: 0)
            env.step_policy(a, 0 if clock is not None else t, pol, nxt, shaped_out=sh, sum_r_out=sr, collision_out=co,
                            clock=clock, seed_offset=0, want_chobs=want_chobs, stuck_penalty=pen if sh is not None else None)
            if vary and t % cfg.episode_interval == cfg.episode_interval - 1:
                env.update_velocity(seed=vel_seed + t // cfg.episode_interval)
            if clock is not None:
                env.lib.diral_clock_add(clock.ptr(), 1, env._stream())
            a, nxt = nxt, a
            t += 1
        sh0 = torch.zeros((B, N), dtype=dt, device="cuda:0")
        for _ in range(t0):
            one(sh0 if pen else None)
        outs = []
        for rep in range(2):
            sh = torch.zeros((K, B, N), dtype=dt, device="cuda:0")
            sr = torch.zeros((K, B), dtype=dt, device="cuda:0")
            co = torch.zeros((K, B), dtype=dt, device="cuda:0")
            if fused_k:
                env.step_policy(a, 0 if clock is not None else t, pol, nxt, shaped_out=sh, sum_r_out=sr, collision_out=co, slots=K,
                                vel_seed=vel_seed, clock=clock, seed_offset=0, want_chobs=want_chobs, stuck_penalty=pen)
                assert env.last_kernel() & KERNEL_POLICY
                if clock is not None:
                    env.lib.diral_clock_add(clock.ptr(), K, env._stream())
                a, nxt = nxt, a
                t += K
            else:
                for k in range(K):
                    one(sh[k], sr[k], co[k])
            outs.append((sh, sr, co, env._obs.clone(), env._rew.clone(), env._done.clone(), a.clone(),
                         env._chobs.clone() if want_chobs else None))
        torch.cuda.synchronize()
        runs.append((env, pol, outs, pen))
    (e1, p1, o1, pen1), (e2, p2, o2, pen2) = runs
    for rep in range(2):
        for i, name in enumerate(("shaped", "sum_r", "collisions", "state", "reward", "done", "actions")):
            assert torch.equal(o1[rep][i], o2[rep][i]), (rep, name, (o1[rep][i] != o2[rep][i]).nonzero()[:4])
        if want_chobs:
            assert torch.equal(o1[rep][7], o2[rep][7]), rep
    assert torch.equal(p1.prev_action, p2.prev_action) and torch.equal(p1.counter, p2.counter)
    sa, sb = e1.export_state(), e2.export_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(e1.metrics(), e2.metrics())
    if pen1 is not None:
        assert torch.equal(pen1[2], pen2[2]) and torch.equal(pen1[3], pen2[3]) and int(pen1[2].max()) > 2
        assert float((o1[1][0] == -10.0 + o1[1][1][:, :, None] / N).sum()) > 0           # the penalty branch ran
    if vary:
        vcol = cfg.state_space - 1                                                   # add_velocity: the last column
        assert torch.equal(o1[1][3][:, :, vcol].double(), sa["vel"].to(o1[1][3].dtype).double()) or case == "rich_vel_last_ends"
        assert not torch.equal(sa["vel"], torch.full_like(sa["vel"], 1.7))
    e1.check()
    e2.check()


@pytest.mark.parametrize("N,A", [(128, 64), (100, 16)])
def test_flagged_passes_whose_far_entries_stay_put_run_coded_and_propagation_falls_back(N, A, monkeypatch):
    """The far-entry guard of the packed form at N <= 128 (csrc/step_wide.hpp `wide_far_guard`): a highway that is one
    cluster for a dozen slots, then two clusters 2 km apart (positions imported) - every entry about the other cluster ages
    beyond the codes, every pass is flagged, and no far value can move: the guard sends those passes down the coded path.
    Then single vehicles are moved into the gap as relays (stale stamps start to travel from viewer to viewer: the guard
    must refuse and the chain pass run), then everything is one cluster again.  State, reward, channel observation every
    slot and the exported tables at every phase change against the oracle, bit for bit."""
    from oracle.oracle import Oracle, SQ_IEEE
    from diral_amd.config import KERNEL_PACKED, KERNEL_WIDE
    B, L = 6, 4000.0
    monkeypatch.setenv("DIRAL_TABLE_FORM", "packed")               # (N = 100 on 4 km is below the density the packed form is chosen at)
    cfg = bench_config(N, A, L).replace(track_arrival=True)
    rng = np.random.default_rng(77 + N)
    x0 = rng.uniform(1000.0, 1400.0, size=(B, N))
    v0 = rng.uniform(1.1, 1.3, size=(B, N))
    env = make_env(cfg, B)
    env.reset_topology(x0, np.zeros((B, N)), v0)
    orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE, threads=4)
    orc.reset(x0, np.zeros((B, N)), v0)

    def run(t0, t1):
        for t in range(t0, t1):
            a = rng.integers(0, A, size=(B, N)).astype(np.int32)
            obs, rew, chobs, _ = gpu_step(env, STEP_MY_STEP, a, t)
            o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
            o_state = orc.obtain_state(a, o_chobs, o_rew)
            assert np.array_equal(rew, o_rew) and np.array_equal(chobs, o_chobs), t
            assert np.array_equal(obs, o_state), (t, np.argwhere(obs != o_state)[:4])
        assert (env.last_kernel() & 15) == KERNEL_WIDE and (env.last_kernel() & KERNEL_PACKED)

    def tables_equal(tag):
        st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        oe = orc.export()
        for k in ("pos_x", "seq", "x"):
            assert np.array_equal(st[k], oe[k]), (tag, k)
        assert np.array_equal(st["age"], np.minimum(oe["age"], 255)), tag
        return oe

    def move(fn):
        oe = orc.export()
        px = oe["pos_x"].copy()
        fn(px)
        env.import_state(pos_x=px)
        orc.import_state(pos_x=px)

    run(0, 12)                                                     # one cluster: everybody hears everybody, codes only
    half = N // 2
    move(lambda px: px.__setitem__((slice(None), slice(half, None)), px[:, half:] + 2000.0))
    run(12, 60)                                                    # two clusters: the entries across age beyond the codes and stay put
    oe = tables_equal("split")
    own = np.einsum("bkk->bk", oe["seq"])
    assert ((own[:, None, :] - oe["seq"]) >= 8).mean() > 0.3       # ... nearly half of all entries by now
    # relays: a few vehicles of the first cluster in the gap, 240 m apart - stale stamps about the far cluster start to travel
    def relays(px):
        for i in range(8):
            px[:, i] = 1500.0 + 230.0 * i
    move(relays)
    run(60, 90)
    tables_equal("relays")
    move(lambda px: px.__setitem__((slice(None), slice(half, None)), px[:, half:] - 2000.0))
    run(90, 120)                                                   # one cluster again: everything comes back within the codes
    oe = tables_equal("merged")
    env.check()


def test_a_handle_that_dies_inside_a_stream_capture_is_destroyed_later():
    """`diral_env_destroy` frees device memory; a finalizer that the cyclic collector runs between two captured launches
    of ANOTHER handle would take the process down (hipFree inside a capture).  VecV2VEnv.close parks such a handle and
    destroys it with the next close outside a capture."""
    import gc
    from diral_amd import vec_env
    cfg = c2_config()
    victim = make_env(cfg, 4, dtype=torch.float32)
    live = make_env(cfg, 4, dtype=torch.float32)
    for e in (victim, live):
        e.reset_topology(seed=2)
    acts = live.sample(seed=1)
    live.step(acts, 0)
    torch.cuda.synchronize()
    parked_before = len(vec_env._PARKED)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            live.step(acts, 1)
            del victim                                              # dies here: inside the capture
            gc.collect()
            live.step(acts, 2)
    assert len(vec_env._PARKED) == parked_before + 1
    g.replay()
    torch.cuda.synchronize()
    live.check()
    live.close()                                                    # outside a capture: the parked handle goes with it
    assert len(vec_env._PARKED) == 0
