"""CPU-only tests of the host side: C-ABI surface, config mirror, the
reference-shaped TestEnv shim, sharding and the metric reduction over gloo."""
import ctypes
import os
import re

import numpy as np
import pytest

from diral_amd import _lib
from diral_amd.compat import TestEnv
from diral_amd.config import (ConfigError, DiralCfg, EnvConfig, M_COLUMNS, bench_config, c2_config)
from diral_amd.shard import env_shard
from tests.golden_util import Golden, golden_names
from tests.oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C-ABI ----------------------------------------------------------------------

def header_functions():
    src = open(os.path.join(ROOT, "include", "diral_env.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(diral_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 19
    assert sorted(_lib.SYMBOLS) == names, "diral_amd/_lib.py and include/diral_env.h disagree"
    for n in names:
        assert hasattr(lib, n), n
    from diral_amd.config import ABI_VERSION
    src = open(os.path.join(ROOT, "include", "diral_env.h")).read()
    assert lib.diral_env_abi_version() == ABI_VERSION == int(re.search(r"#define DIRAL_ABI_VERSION (\d+)", src).group(1)) == 8


def test_cfg_struct_layout_matches_header():
    lib = _lib.load()
    c = DiralCfg()
    lib.diral_cfg_defaults(ctypes.byref(c))
    assert c.struct_bytes == ctypes.sizeof(DiralCfg) == 88
    # the reference's kwargs.setdefault defaults (test_env.py:12-24)
    assert (c.num_users, c.num_channels, c.reward_design) == (3, 3, 1)
    assert (c.highway_length, c.communication_range, c.bin_range) == (200.0, 1.0, 500.0)
    assert c.info_age_limit == 20 and c.episode_interval == 25 and c.pf_threshold == 10 and c.pf_penalty == -10.0


def test_slot_policy_struct_layout_and_argument_checks():
    """`DiralSlotPolicy` (diral_env_step_policy): the ctypes image has the header's size (a gcc-compiled probe of the
    header), and the entry point rejects null handles / wrong struct sizes without touching a device."""
    import subprocess
    import tempfile
    from diral_amd.config import DiralSlotPolicy
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        with open(c, "w") as fh:
            fh.write('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) { printf("%%zu %%zu %%zu %%zu\\n", '
                     'sizeof(DiralSlotPolicy), offsetof(DiralSlotPolicy, shaped_out), offsetof(DiralSlotPolicy, rssi_threshold), '
                     'offsetof(DiralSlotPolicy, actions_out)); return 0; }\n' % os.path.join(ROOT, "include", "diral_env.h"))
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", c, "-o", exe])
        size, o1, o2, o3 = (int(x) for x in subprocess.check_output([exe]).split())
    assert size == ctypes.sizeof(DiralSlotPolicy)
    assert (o1, o2, o3) == (DiralSlotPolicy.shaped_out.offset, DiralSlotPolicy.rssi_threshold.offset, DiralSlotPolicy.actions_out.offset)
    lib = _lib.load()
    q = DiralSlotPolicy()
    q.struct_bytes = ctypes.sizeof(DiralSlotPolicy)
    assert lib.diral_env_step_policy(None, 0, None, 0, None, None, None, None, 0, ctypes.byref(q), None) == -1


def test_host_validation_and_state_space():
    lib = _lib.load()
    for name in golden_names():
        g = Golden(name)
        c = g.cfg.to_c()
        assert lib.diral_env_state_space(ctypes.byref(c)) == int(g["state_space"]) == g.cfg.state_space, name
    ok = c2_config().to_c()
    assert lib.diral_env_validate(ctypes.byref(ok)) == 0
    bad = c2_config().to_c(); bad.reward_design = 9
    assert lib.diral_env_validate(ctypes.byref(bad)) == -2
    # sizes: the reference has no limits (test_env.py:12-13, 40); beyond 256 vehicles / 256 resources / 64 bins the
    # three-launch form (csrc/step_large.hpp) runs, up to 4096 / 4096 / 1024 while one env fits a workgroup's LDS
    for n, a, k, want in ((257, 32, 20, 0), (4096, 4096, 20, 0), (4097, 32, 20, -3), (64, 4097, 20, -3), (64, 32, 1024, 0),
                          (64, 32, 1025, -3), (200, 200, 64, 0)):
        big = c2_config().to_c(); big.num_users = n; big.num_channels = a; big.num_bins = k
        assert lib.diral_env_validate(ctypes.byref(big)) == want, (n, a, k)
    pb = c2_config(State=dict(piggybacking=True, add_channel_obs=True)).to_c()
    assert lib.diral_env_validate(ctypes.byref(pb)) == 0
    pb.num_users = 300                                        # State.piggybacking: any number of vehicles, up to 256 resources
    assert lib.diral_env_validate(ctypes.byref(pb)) == 0
    pb.num_channels = 257
    assert lib.diral_env_validate(ctypes.byref(pb)) == -3
    short = c2_config().to_c(); short.struct_bytes = 4
    assert lib.diral_env_validate(ctypes.byref(short)) == -1
    assert b"action" in lib.diral_env_strerror(-6)
    # null handles are rejected, not dereferenced
    assert lib.diral_env_step(None, 0, None, 0, None, None, None, None, 0, 0.0, 1.0, None) == -1
    assert lib.diral_env_destroy(None) == 0


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = ctypes.c_void_p()
    c = c2_config().to_c()
    assert lib.diral_env_create(ctypes.byref(c), 4, 0, ctypes.byref(h)) == -5
    from diral_amd.vec_env import DiralError, VecV2VEnv
    with pytest.raises(DiralError):
        VecV2VEnv(c2_config(), batch=2)        # no silent CPU fallback


# ---- config mirror --------------------------------------------------------------

def test_config_accepts_reference_dict_verbatim():
    g = Golden("g1_step_rd2")
    cfg = EnvConfig.from_dict(g.cfg_dict)
    assert (cfg.num_users, cfg.num_channels, cfg.State.num_bins, cfg.state_space) == (4, 3, 20, 23)
    assert cfg.congestion_test and cfg.mobility and not cfg.mobility_vary
    assert cfg.extra == {}
    with pytest.raises(ConfigError):
        EnvConfig.from_dict({"num_users": 4})                     # reference: State missing -> TypeError
    d = dict(g.cfg_dict); d["State"] = dict(d["State"]); del d["State"]["num_bins"]
    with pytest.raises(ConfigError):
        EnvConfig.from_dict(d)                                    # reference: KeyError
    # State.piggybacking (test_env.py:71-79, 241-254): defined with State.type 2 + add_channel_obs only
    for bad in (dict(reward_design=0), dict(State=dict(piggybacking=True)), dict(State=dict(type=3)),
                dict(State=dict(piggybacking=True, add_channel_obs=True, type=1)),
                dict(mobility=False), dict(State=dict(action_index="hex"))):
        with pytest.raises(ConfigError):
            bench_config(8, 4, 100.0, **bad).validate()
    pb = bench_config(8, 4, 100.0, State=dict(piggybacking=True, add_channel_obs=True))
    pb.validate()
    assert pb.chobs_width == 16 and pb.state_space == 4 + 4 + 4 * 3 + 20        # test_env.py:49-85
    from diral_amd.config import F_PIGGYBACKING
    assert pb.to_c().flags & F_PIGGYBACKING


def test_piggybacking_fixture_config_and_the_oracle_backed_shim():
    """State.piggybacking through the reference-shaped shim on the CPU-backed test backend: obs[user] holds A * A
    values, the state vector get_state_space() columns (g1_piggyback: 32), and the slot in which a receiver hears
    nobody raises KeyError like the reference (test_env.py:243)."""
    from diral_amd.compat import TestEnv
    from tests.oracle_backend import OracleBackend
    g = Golden("g1_piggyback")
    assert g.cfg.State.piggybacking and g.cfg.state_space == int(g["state_space"]) == 32
    env = TestEnv(backend=OracleBackend(g.cfg), **g.cfg_dict)
    env.reset_mobility_env()
    for i, mode, acts, t, (ep, eps) in g.steps():
        obs, rews = env.my_step(acts, t)
        assert sorted(obs) == list(range(4)) and all(obs[u].shape == (9,) for u in obs)
        st = env.obtain_state(obs, acts, rews, ep, eps)
        assert np.array_equal(np.array([obs[u] for u in range(4)]), g["chobs"][i])
        assert np.array_equal(np.array(st), g["state"][i])
    gk = Golden("g8_piggyback_keyerror")
    envk = TestEnv(backend=OracleBackend(gk.cfg), **gk.cfg_dict)
    envk._env.reset_topology(gk["x0"], gk["y0"], gk["v0"])
    for i, mode, acts, t, _ in gk.steps():
        envk.my_step(acts, t)
    with pytest.raises(KeyError):
        envk.my_step(gk["keyerror_actions"], int(gk["keyerror_t"]))


def test_config_from_yaml(tmp_path):
    import yaml
    g = Golden("g1_step_rd2")
    doc = {"experiment_name": "x", "episode_interval": 25, "EnvironmentTest": g.cfg_dict, "RLAgent": {"gamma": 0.7}}
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(doc))
    cfg = EnvConfig.from_yaml(str(p))
    assert cfg.state_space == 23 and cfg.episode_interval == 25 and cfg.communication_range == 250


# ---- reference-shaped shim (main_test.py call sequence) -----------------------

@pytest.mark.parametrize("name", ["g1_step_rd2", "g1_ch_rd3", "g1_design", "g1_flags_all"])
def test_testenv_shim_reproduces_reference_shapes_and_values(name):
    g = Golden(name)
    env = TestEnv(backend=OracleBackend(g.cfg), **g.cfg_dict)
    env.reset_mobility_env()
    assert env.get_total_users() == 4 and env.get_action_space() == 3 and env.get_num_ch() == 3
    assert env.get_state_space() == int(g["state_space"])
    step = {0: env.my_step, 1: env.my_step_ch, 2: env.my_step_design}
    for i, mode, acts, t, (ep, eps) in g.steps():
        obs, rews = step[mode](acts, t)
        assert isinstance(obs, dict) and sorted(obs) == [0, 1, 2, 3] and obs[0].shape == (3,)
        assert isinstance(rews, np.ndarray) and rews.shape == (4,)
        state = env.obtain_state(obs, acts, list(rews), ep, eps)
        assert isinstance(state, list) and len(state) == 4 and state[0].shape == (env.get_state_space(),)
        assert np.array_equal(np.array(state), g["state"][i])
        assert np.array_equal(rews, g["rews"][i])
        assert np.array_equal(np.array([obs[u] for u in range(4)]), g["chobs"][i])
        assert env.get_x_pos() == list(g["pos_x"][i])
        assert env.network.get_information_age(t) == list(g["ia"][i])
    a = env.sample()
    assert a.shape == (4,) and a.min() >= 0 and a.max() < 3
    assert list(env.one_hot(2, 3)) == [0, 0, 1]


# ---- sharding + metric reduction (world_size 2, gloo) --------------------------

def test_env_shard_partitions():
    for total in (0, 1, 7, 4096, 262144):
        for world in (1, 2, 3, 8):
            spans = [env_shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert env_shard(262144, 3, 8) == (98304, 32768)              # BASELINE configs[3]
    with pytest.raises(ValueError):
        env_shard(8, 2, 2)


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from diral_amd.metrics import reduce_metric_sums, summarize
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = env_shard(10, rank, world)
    local = torch.zeros((count, M_COLUMNS), dtype=torch.float64)
    for j in range(count):
        e = start + j
        local[j] = torch.tensor([25.0, -3.0 * e, 10.0 + e, 54.0 - e, 40.0 + e, 64.0])
    packed = reduce_metric_sums(local)
    q.put((rank, packed.tolist(), summarize(packed, 32)))
    dist.barrier()
    dist.destroy_process_group()


def test_metric_allreduce_over_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    e = np.arange(10)
    want = [250.0, float(-3.0 * e.sum()), float((10 + e).sum()), float((54 - e).sum()), float((40 + e).sum()),
            640.0, 10.0]
    for rank, packed, summ in res:
        assert packed == want, (rank, packed)
        assert summ["envs"] == 10 and summ["env_slots"] == 250
        assert abs(summ["prr"] - (40 + e).sum() / 640.0) < 1e-12
        assert abs(summ["collision_fraction"] - (54 - e).sum() / 640.0) < 1e-12


def test_byte_models_of_the_roofline_bookkeeping():
    """SURVEY 8d's canonical bytes per env-slot (reported as `roofline.model_*`) at the three bench configurations,
    and this build's layout bytes (`roofline.achieved` / `frac`): packed code + age bytes (the BASELINE configurations),
    4-byte (seq, age) words for step_wide's plane form, plus the subjects' xpos rings."""
    from diral_amd.roofline import (algorithmic_bytes_per_env_slot as alg, layout_bytes_per_env_slot as lay,
                                    packed_table)
    assert alg(64, 32, 52) == 155136 and alg(256, 64, 84) == 2258944 and alg(128, 64, 84) == 605184
    # the form a handle takes by density (csrc/diral_env.hip use_packed_table): BASELINE configs[1], [2], [4] packed,
    # sparse highways at N > 64 on the (seq, age) plane
    assert packed_table(64, 250, 2000) and packed_table(256, 250, 4000) and packed_table(128, 250, 4000)
    assert packed_table(64, 10, 9000) and not packed_table(128, 250, 5000) and not packed_table(256, 250, 8000)
    with pytest.raises(TypeError):
        lay(128, 64, 84, False)                    # the form is the caller's to state
    # entries read + written, ring rows read + one stamp written (+ own sequence numbers r/w), per-vehicle arrays,
    # reward, state (+ channel observation)
    assert lay(64, 32, 52, False, packed=True) == 2 * 2 * 64 * 64 + 80 * 64 + 36 * 64 + 4 * 64 + 4 * 64 * 52
    assert lay(64, 32, 52, True, packed=True) - lay(64, 32, 52, False, packed=True) == 4 * 64 * 32
    assert lay(128, 64, 84, False, packed=False) == 2 * 4 * 128 * 128 + 72 * 128 + 36 * 128 + 4 * 128 + 4 * 128 * 84
    assert lay(128, 64, 84, False, packed=True) == 2 * 2 * 128 * 128 + 80 * 128 + 36 * 128 + 4 * 128 + 4 * 128 * 84
    assert lay(256, 64, 84, False, packed=True) == 2 * 2 * 256 * 256 + 80 * 256 + 36 * 256 + 4 * 256 + 4 * 256 * 84
    assert 5 * lay(256, 64, 84, False, packed=True) < alg(256, 64, 84)
    # where a launch's reads can come from (`roofline.memory`): the state C2's 4096 envs re-read every launch fits the
    # 256 MiB Infinity Cache (the counters then see fabric bytes), the C4 shard's and C3's do not
    from diral_amd.roofline import INFINITY_CACHE_BYTES, memory_level, resident_bytes_per_env
    assert resident_bytes_per_env(64, packed=True) == 2 * 64 * 64 + 64 * 64 + 4 * 64 + 28 * 64
    m2, m4, m3 = (memory_level(64, 32, 52, 4096, True, packed=True), memory_level(64, 32, 52, 32768, True, packed=True),
                  memory_level(256, 64, 84, 8192, True, packed=True))
    assert m2["resident_bytes"] < INFINITY_CACHE_BYTES < m4["resident_bytes"] < m3["resident_bytes"]
    assert m2["reads_served_by"].startswith("infinity cache") and m4["reads_served_by"].startswith("hbm")
    assert m2["output_bytes"] == 4096 * (4 * 64 + 4 * 64 * 52 + 4 * 64 * 32)
