"""the driver-loop harness (diral_amd/driver.py)
against fixtures recorded from the reference env under main_test.py's own call
sequence (tests/golden/gen_golden.py::run_driver_case).  CPU run: oracle-backed
env stand-in, bit-exact; GPU run: the real VecV2VEnv."""
import json
import os

import numpy as np
import pytest
import torch

from diral_amd.config import EnvConfig
from diral_amd.driver import DriverLoop, calculate_ia_penalty, np_sum_lastdim
from tests.golden_util import GOLDEN_DIR, golden_names, ulp_diff
from tests.oracle_backend import OracleBackend


def load(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = EnvConfig.from_dict(json.loads(str(d["cfg"])), track_arrival=True)
    return d, cfg, json.loads(str(d["opts"]))


def run(env, d, opts, exact, fused=False):
    loop = DriverLoop(env, fused=fused, enable_channel=opts["enable_channel"], global_reward_avg=opts["global_reward_avg"],
                      ia_averaging=opts["ia_averaging"], ia_penalty_enable=opts["ia_penalty_enable"],
                      ia_penalty_threshold=opts["ia_penalty_threshold"], ia_penalty_value=opts["ia_penalty_value"],
                      episode_interval=opts["episode_interval"])
    env.reset_topology(d["x0"], d["y0"], d["v0"])

    def np_(t):
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def same(a, b, what):
        a = np_(a).astype(np.float64)
        if exact:
            assert np.array_equal(a, b), what
        else:
            assert ulp_diff(a, b) <= 1 or np.allclose(a, b, rtol=0, atol=4e-15), what

    st = loop.bootstrap(d["boot_action"])
    same(st[0], d["boot_state"], "bootstrap state")
    for i in range(opts["n_prefill"]):
        st = loop.prefill_step(d["pre_actions"][i])
        same(st[0], d["pre_states"][i], ("prefill", i))
    for t in range(opts["T"]):
        assert loop.episode == (0 if t == 0 else int(d["episode"][t - 1]))
        out = loop.slot(d["actions"][t], t, want_ia=True)
        same(out["next_state"][0], d["states"][t], ("state", t))
        same(out["raw_reward"][0], d["raw_reward"][t], ("raw reward", t))
        same(out["reward"][0], d["shaped_reward"][t], ("shaped reward", t))
        same(out["sum_r"][0], d["sum_r"][t], ("sum_r", t))
        same(out["collision"][0], d["collision"][t], ("collision", t))
        assert np.array_equal(np_(out["ia"])[0], d["ia"][t]), ("ia", t)
        assert int(np_(out["ia_sum"])[0]) == int(d["ia_sum"][t])
        if opts["ia_averaging"]:
            assert int(np_(out["ia_penalty"])[0]) == int(d["ia_pen"][t])
        assert bool(out["episode_end"]) == bool(d["episode_end"][t])
        if out["episode_end"]:
            loop.end_episode(d["vel_draws"][t])
        assert abs(loop.eps - float(d["eps"][t])) < 1e-15 and loop.episode == int(d["episode"][t])
    fin = env.export_state()
    assert np.array_equal(np_(fin["pos_x"])[0], d["final_pos"])
    assert np.array_equal(np_(fin["vel"])[0], d["final_vel"])
    assert np.array_equal(np_(fin["seq"])[0], d["final_seq"])
    assert np.array_equal(np_(fin["x"])[0], d["final_x"])


@pytest.mark.parametrize("name", golden_names("d"))
def test_driver_loop_reproduces_reference_sequence_cpu(name):
    d, cfg, opts = load(name)
    run(OracleBackend(cfg), d, opts, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names("d"))
def test_driver_loop_reproduces_reference_sequence_gpu(name):
    from diral_amd.vec_env import VecV2VEnv
    d, cfg, opts = load(name)
    run(VecV2VEnv(cfg, batch=1, out_dtype=torch.float64), d, opts, exact=False)
    if not cfg.State.add_reward:       # one fused launch per slot gives the same sequence
        run(VecV2VEnv(cfg, batch=1, out_dtype=torch.float64), d, opts, exact=False, fused=True)


def test_ia_penalty_helper_matches_utils_misc():
    rng = np.random.default_rng(0)
    ia = rng.integers(0, 9, size=(7, 100))
    want = [sum((i + 1) * v for i, v in enumerate(row) if v > 0) for row in ia]      # utils/misc.py:1-12
    assert calculate_ia_penalty(torch.as_tensor(ia)).tolist() == want


def test_np_sum_order_matches_numpy():
    rng = np.random.default_rng(1)
    for n in (1, 3, 7, 8, 9, 16, 63, 64, 100, 128, 129, 200, 256, 1000):
        a = rng.normal(size=(5, n)) * 10.0 ** rng.integers(-3, 4, size=(5, n))
        got = np_sum_lastdim(torch.as_tensor(a)).numpy()
        want = np.array([np.sum(row) for row in a])
        assert np.array_equal(got, want), n


@pytest.mark.gpu
@pytest.mark.parametrize("N,dtype", [(4, torch.float64), (7, torch.float32), (16, torch.float64), (64, torch.float32),
                                     (64, torch.float64), (129, torch.float64), (200, torch.float32), (256, torch.float64)])
def test_device_reward_shaping_equals_the_torch_statement(N, dtype):
    """`diral_driver_shape` (main_test.py:150-206 in one launch) against the torch statement of
    DriverLoop.slot on the same env: shaped rewards, np.sum-ordered sum_r, collision metric,
    information-age sum and penalty, stuck-action counters - bit for bit, float32 and float64,
    every option on."""
    from diral_amd.config import bench_config
    from diral_amd.vec_env import VecV2VEnv
    A = 8
    cfg = bench_config(N, A, 30.0 * N + 50, reward_design=3, communication_range=120.0).replace(track_arrival=True)
    B = 24
    envs = [VecV2VEnv(cfg, batch=B, out_dtype=dtype) for _ in range(2)]
    loops = [DriverLoop(e, enable_channel=True, global_reward_avg=True, ia_averaging=True, ia_penalty_enable=True,
                        ia_penalty_threshold=2, ia_penalty_value=-10, device_shaping=ds) for e, ds in zip(envs, (True, False))]
    assert loops[0].device_shaping and not loops[1].device_shaping
    for e in envs:
        e.reset_topology(seed=31)
    rng = np.random.default_rng(N)
    acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
    for lp in loops:
        lp.bootstrap(torch.as_tensor(acts, device="cuda:0"))
    for t in range(40):
        new = rng.integers(0, A, size=(B, N))
        acts = np.where(rng.random((B, N)) < 0.7, acts, new).astype(np.int32)       # sticky: the stuck counters fire
        a = torch.as_tensor(acts, device="cuda:0")
        o_dev, o_ref = loops[0].slot(a, t, want_ia=True), loops[1].slot(a, t, want_ia=True)
        for k in ("reward", "raw_reward", "sum_r", "collision", "ia", "ia_sum", "ia_penalty", "next_state"):
            assert torch.equal(o_dev[k].to(o_ref[k].dtype), o_ref[k]), (k, t)
        assert o_dev["reward"].dtype == dtype and o_dev["sum_r"].dtype == dtype
        if o_dev["episode_end"]:
            for lp in loops:
                lp.end_episode(np.full((B, N), 3, np.uint8))
    assert torch.equal(loops[0]._pen_counter.long(), loops[1]._pen_counter.long())
    assert int(loops[0]._pen_counter.max()) > 2          # the threshold penalty really fired


@pytest.mark.gpu
@pytest.mark.parametrize("N,dtype,pen", [(8, torch.float64, True), (9, torch.float32, False), (15, torch.float64, True),
                                         (16, torch.float32, True), (40, torch.float64, False), (63, torch.float32, True),
                                         (64, torch.float32, False), (64, torch.float64, True), (7, torch.float64, True)])
def test_device_reward_shaping_wave_kernel_equals_the_torch_statement(N, dtype, pen):
    """`diral_driver_shape` without information-age terms takes the one-wave-per-env kernel at 8 <= N <= 64
    (np.sum's pairwise order walked with lane shuffles): against the torch statement, bit for bit."""
    from diral_amd.config import bench_config
    from diral_amd.vec_env import VecV2VEnv
    A = 6
    cfg = bench_config(N, A, 30.0 * N + 50, reward_design=3, communication_range=120.0)
    B = 37
    envs = [VecV2VEnv(cfg, batch=B, out_dtype=dtype) for _ in range(2)]
    loops = [DriverLoop(e, global_reward_avg=True, ia_penalty_enable=pen, ia_penalty_threshold=2, ia_penalty_value=-10,
                        device_shaping=ds) for e, ds in zip(envs, (True, False))]
    for e in envs:
        e.reset_topology(seed=31)
    rng = np.random.default_rng(N)
    acts = rng.integers(0, A, size=(B, N)).astype(np.int32)
    for lp in loops:
        lp.bootstrap(torch.as_tensor(acts, device="cuda:0"))
    for t in range(30):
        new = rng.integers(0, A, size=(B, N))
        acts = np.where(rng.random((B, N)) < 0.7, acts, new).astype(np.int32)
        a = torch.as_tensor(acts, device="cuda:0")
        o_dev, o_ref = loops[0].slot(a, t), loops[1].slot(a, t)
        for k in ("reward", "raw_reward", "sum_r", "collision", "next_state"):
            assert torch.equal(o_dev[k].to(o_ref[k].dtype), o_ref[k]), (k, t)
    if pen:
        assert torch.equal(loops[0]._pen_counter.long(), loops[1]._pen_counter.long())


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_io_ring_keeps_the_previous_slot_intact_without_copies(fused):
    """VecV2VEnv(io_ring=2): the step calls alternate between two output-buffer sets, DriverLoop
    makes no defensive copies, and (state, next_state) of consecutive slots (what memory.add takes,
    main_test.py:207-214) are still two intact tensors; same values as the copying loop."""
    from diral_amd.config import c2_config
    from diral_amd.vec_env import VecV2VEnv
    cfg = c2_config()
    B, N, A = 12, 64, 32
    e1, e2 = VecV2VEnv(cfg, batch=B), VecV2VEnv(cfg, batch=B, io_ring=2)
    l1, l2 = (DriverLoop(e, global_reward_avg=True, fused=fused) for e in (e1, e2))
    assert l1._own_outputs and not l2._own_outputs
    for e in (e1, e2):
        e.reset_topology(seed=77)
    rng = np.random.default_rng(3)
    a0 = torch.as_tensor(rng.integers(0, A, size=(B, N)).astype(np.int32), device="cuda:0")
    s1, s2 = l1.bootstrap(a0), l2.bootstrap(a0)
    prev = None
    for t in range(30):
        a = torch.as_tensor(rng.integers(0, A, size=(B, N)).astype(np.int32), device="cuda:0")
        o1, o2 = l1.slot(a, t), l2.slot(a, t)
        for k in ("next_state", "reward", "raw_reward", "sum_r", "collision"):
            assert torch.equal(o1[k], o2[k]), (k, t)
        if prev is not None:
            # the previous slot of the ring env: still what the copying loop holds
            assert o2["next_state"].data_ptr() != prev[1]["next_state"].data_ptr()
            assert torch.equal(prev[0]["next_state"], prev[1]["next_state"]), t
            assert torch.equal(prev[0]["raw_reward"], prev[1]["raw_reward"]), t
        prev = (o1, o2)
    with pytest.raises(ValueError):
        VecV2VEnv(cfg, batch=1, io_ring=0)


def test_driver_loop_prefill_as_a_whole_equals_its_steps_cpu():
    """DriverLoop.prefill (main_test.py:99-114 as one call: K random slots, every state and action kept) on an env
    without the one-launch form - the oracle-backed stand-in - is the loop of `prefill_step` calls on the draws
    `sample(seed + k)` makes; on the HIP env it is one launch, compared with this loop in tests/test_gpu_prefill.py."""
    from diral_amd.config import c2_config
    cfg = c2_config(State=dict(add_reward=True, add_channel_obs=True))
    for ch in (False, True):
        envs = [OracleBackend(cfg, batch=2), OracleBackend(cfg, batch=2)]
        loops = [DriverLoop(e, enable_channel=ch) for e in envs]
        for e, lp in zip(envs, loops):
            e.reset_topology(seed=3)
            lp.bootstrap(e.sample(11))
        states, acts = loops[0].prefill(6, 500)
        assert tuple(states.shape) == (6, 2, cfg.num_users, cfg.state_space) and tuple(acts.shape) == (6, 2, cfg.num_users)
        for k in range(6):
            a = envs[1].sample(500 + k)
            assert np.array_equal(acts[k].numpy(), a)
            assert np.array_equal(states[k].numpy(), loops[1].prefill_step(a).numpy()), (ch, k)
        assert np.array_equal(envs[0].export_state()["seq"], envs[1].export_state()["seq"])
