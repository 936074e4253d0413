"""Full-size parity of the BENCHMARKED instantiations (BASELINE.json configs[1], [2], [4]): the exact kernel,
output dtype and outputs bench.py times - f32 state + reward + channel observation from ONE launch, device-drawn
topology from the bench's seed - at the bench's batch sizes.  Size-independent properties on EVERY env, and
sampled envs compared bit for bit with the oracle (state, reward AND channel observation; the f32 outputs are
the float32 cast of the reference's float64 values)."""
import os

import numpy as np
import pytest
import torch

from diral_amd.config import (KERNEL_FAST64, KERNEL_PACKED, KERNEL_RICH, KERNEL_RING, KERNEL_WIDE, STEP_MY_STEP,
                              bench_config)

pytestmark = pytest.mark.gpu

GLOBAL_SEED = 1234      # bench.py's


def _run(N, A, L, B, vary, n_sample, T, fam, vel_slot=None, dtype=torch.float32, env_offset=0, n_gap=0, min_slow=0, min_dense=8):
    from diral_amd.vec_env import VecV2VEnv
    from oracle.oracle import Oracle, SQ_IEEE
    cfg = bench_config(N, A, L, mobility_vary=vary)
    K = cfg.State.num_bins
    npdt = np.float32 if dtype == torch.float32 else np.float64
    env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=dtype, env_offset=env_offset)
    env.reset_topology(seed=GLOBAL_SEED)                    # the device draws bench.py uses
    st0 = env.export_state(tables=False)
    rng = np.random.default_rng(N * 1000 + A)
    sample = rng.choice(B, size=n_sample, replace=False)
    if n_gap:
        # ... and the envs the kernel's scheduler treats specially: a highway with a gap wider than the communication range
        # keeps the tables on either side stale - entries beyond the 8-level codes, quads on the keyed path, the env on the
        # slow-first list (csrc/step_fast64.hpp).  The n_gap widest gaps of the batch (no wrap-around: network.py:318-332).
        xs = torch.sort(st0["pos_x"], dim=1).values
        gap = (xs[:, 1:] - xs[:, :-1]).max(dim=1).values
        widest = torch.argsort(gap, descending=True)[:n_gap].cpu().numpy()
        assert float(gap[int(widest[-1])]) > cfg.communication_range, "the batch holds no %d envs with a gap > Rc" % n_gap
        sample = np.concatenate([sample, widest])
    sample = np.unique(sample)
    n_sample = len(sample)
    sample_t = torch.as_tensor(sample, device="cuda:0")
    orc = Oracle(cfg, batch=n_sample, sq_mode=SQ_IEEE, threads=8)
    orc.reset(st0["pos_x"][sample_t].cpu().numpy(), st0["pos_y"][sample_t].cpu().numpy(), st0["vel"][sample_t].cpu().numpy())
    g = torch.Generator(device="cuda:0").manual_seed(7)
    packed = os.environ.get("DIRAL_TABLE_FORM") != "plane"       # (the BASELINE configurations are dense: packed by default)
    for t in range(T):
        a_t = torch.randint(0, A, (B, N), device="cuda:0", dtype=torch.int32, generator=g)
        obs, rew, done = env._step(STEP_MY_STEP, a_t, t, want_chobs=True)     # what bench.py's timed step calls
        chobs = env._chobs
        assert env.last_kernel() == fam | KERNEL_RICH | KERNEL_RING | (KERNEL_PACKED if packed else 0), env.last_kernel()
        al = a_t.long()
        # (1) one-hot section == actions
        assert torch.equal(obs[..., :A].argmax(-1), al) and torch.all(obs[..., :A].sum(-1) == 1)
        # (2) histogram rows sum to 1 (or are all zero); every bin is a multiple of 1/n
        hs = obs[..., A:].double().sum(-1)
        assert torch.all(((hs - 1).abs() < 1e-5) | (hs == 0))
        # (3) rewards against collision counts recomputed in torch (reward_design 2)
        cnt = torch.zeros((B, A), dtype=torch.long, device="cuda:0").scatter_add_(1, al, torch.ones_like(al))
        c = cnt.gather(1, al)
        assert torch.all(rew[c == 1] == 1)
        assert torch.all(rew[c > 2] == -c[c > 2].to(dtype))
        assert torch.all((rew[c == 2] == 0) | (rew[c == 2] == -2))
        # (4) channel observation: 0 on the own resource and on unused ones, else a distance < Rc or the 100000 sentinel
        own = torch.gather(chobs, 2, al.unsqueeze(-1))
        assert torch.all(own == 0)
        used = (cnt > 0).unsqueeze(1).expand(B, N, A)
        assert torch.all(chobs[~used] == 0)
        v = chobs[used]
        # (the f32 cast of a float64 distance just below Rc may round up to Rc itself)
        assert torch.all((v == 100000.0) | ((v >= 0) & (v <= cfg.communication_range)))
        assert torch.all(done == (1 if t % cfg.episode_interval == cfg.episode_interval - 1 else 0))
        # (5) sampled envs bit for bit vs the oracle
        acts_s = a_t[sample_t].cpu().numpy()
        o_rew, o_chobs = orc.step(STEP_MY_STEP, acts_s, t)
        o_state = orc.obtain_state(acts_s, o_chobs, o_rew)
        assert np.array_equal(obs[sample_t].cpu().numpy(), o_state.astype(npdt)), t
        assert np.array_equal(rew[sample_t].cpu().numpy(), o_rew.astype(npdt)), t
        assert np.array_equal(chobs[sample_t].cpu().numpy(), o_chobs.astype(npdt)), t
        if vel_slot is not None and t == vel_slot:
            draws = torch.randint(1, 4, (B, N), device="cuda:0", dtype=torch.uint8, generator=g)
            env.update_velocity(draws)
            orc.update_velocity(draws[sample_t].cpu().numpy())
    # (6) tables: own sequence number == slot count, nobody ahead of the subject; sampled envs' planes vs the oracle
    st = env.export_state()
    assert st["pos_x"].min() >= 0 and st["pos_x"].max() < L
    diag = torch.diagonal(st["seq"], dim1=1, dim2=2)
    assert torch.all(diag == T) and torch.all(st["seq"] <= T)
    oe = orc.export()
    for k in ("pos_x", "vel", "seq", "x"):
        assert np.array_equal(st[k][sample_t].cpu().numpy(), oe[k]), k
    assert np.array_equal(st["age"][sample_t].cpu().numpy(), np.minimum(oe["age"], 255))
    if min_slow:
        # which envs ended the run on the slow list: an entry that was heard and lags its subject by 7 stamps or more flags
        # its quad for the keyed path of the next slot (step_fast64.hpp `keep` / `hand`).  Enough of them were compared above.
        seq = st["seq"]
        beyond = ((seq > 0) & (diag.unsqueeze(1) - seq >= 7)).flatten(1).any(dim=1)
        dense = ~((seq == 0).flatten(1).any(dim=1)) & ~beyond          # every entry heard and within the codes: fast quads only
        slow_sampled = int(beyond[sample_t].sum())
        assert slow_sampled >= min_slow, (slow_sampled, int(beyond.sum()))
        assert int(dense[sample_t].sum()) >= min_dense, int(dense[sample_t].sum())
        # slow-first dispatch is on (DIRAL_NO_SLOW_FIRST unset): the listed envs ran in the front-of-grid blocks, every env
        # exactly once - the diagonal check above
    env.check()


def test_c2_benchmarked_instantiation_full_size():
    """configs[1]: 64 UE / 32 res, B = 4096 - step_fast64_kernel<true,false,false,false,true>; 64 slots (past the ghost
    phase of SURVEY Q4 and past lag 7), the 12 envs with the widest highway gaps in the oracle sample beside 48 random ones:
    at least 8 envs that end on the slow-first list and 8 that run on fast quads only are compared bit for bit."""
    _run(64, 32, 2000.0, 4096, False, n_sample=48, T=64, fam=KERNEL_FAST64, n_gap=12, min_slow=8)


def test_c3_benchmarked_instantiation_full_size():
    """configs[2]: 256 UE / 64 res congested, B = 8192 - step_wide_kernel<4,false,true,false,false,true>."""
    _run(256, 64, 4000.0, 8192, False, n_sample=8, T=30, fam=KERNEL_WIDE)


def test_c5_benchmarked_instantiation_full_size():
    """configs[4]: 128 UE / 64 res, mobility_vary, B = 16384 - step_wide_kernel<2,false,true,false,false,true,true>, the packed
    table form; one update_velocity.  One env in ten of this density has broken into clusters that no longer hear each
    other - its passes leave the codes (flagged passes through the planes, byte ranks) and it runs in the front-of-grid
    blocks of the slow-first dispatch: the 6 envs with the widest gaps are in the oracle sample beside 10 random ones, at
    least 4 envs that end the run with an entry beyond the codes are compared bit for bit."""
    _run(128, 64, 4000.0, 16384, True, n_sample=10, T=48, fam=KERNEL_WIDE, vel_slot=24, n_gap=6, min_slow=4, min_dense=0)


def test_c4_shard_benchmarked_instantiation_full_size():
    """configs[3]: one GPU's share of the 262144-env job, B = 32768 at the env offset of rank 5 - the RICH instantiation
    `also_measured.c4shard` times (state + reward + channel observation in one launch), 24 envs sampled across the shard
    against the oracle, 48 slots, with the 12 widest-gap envs of the shard (>= 8 slow-listed ones compared)."""
    _run(64, 32, 2000.0, 32768, False, n_sample=24, T=48, fam=KERNEL_FAST64, env_offset=5 * 32768, n_gap=12, min_slow=8)


@pytest.mark.parametrize("N,A,L,B,vary,ns,T", [(64, 32, 2000.0, 4096, False, 16, 14), (256, 64, 4000.0, 8192, False, 4, 8),
                                               (128, 64, 4000.0, 16384, True, 6, 10)])
def test_float64_outputs_full_size(N, A, L, B, vary, ns, T):
    """The float64-output instantiations (the reference's own array dtype, `also_measured.c2_f64`) at the bench's batch
    sizes: state, reward and channel observation of the sampled envs equal the oracle's float64 values bit for bit."""
    _run(N, A, L, B, vary, n_sample=ns, T=T, fam=KERNEL_FAST64 if N <= 64 else KERNEL_WIDE, dtype=torch.float64)


def test_specialised_kernel_is_well_ahead_of_the_general_one_at_c2():
    """A loose wall-clock check the driver's `-m gpu` run sees (the tight ones live under `-m perf`): at C2 shapes the
    dispatch's kernel is at least 1.2 x as fast as the general kernel forced onto the same handle (measured: > 4 x)."""
    import time
    from diral_amd.vec_env import VecV2VEnv
    cfg = bench_config(64, 32, 2000.0)
    times = {}
    for general in (False, True):
        env = VecV2VEnv(cfg, batch=2048, device="cuda:0", out_dtype=torch.float32)
        env.reset_topology(seed=3)
        env.force_general_kernel(general)
        acts = [env.sample(seed=i) for i in range(4)]
        for t in range(30):
            env.step(acts[t % 4], t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(30, 110):
            env.step(acts[t % 4], t)
        torch.cuda.synchronize()
        times[general] = time.perf_counter() - t0
        env.check()
    assert times[True] >= 1.2 * times[False], times
