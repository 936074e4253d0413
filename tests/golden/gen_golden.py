#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference env.

Runs ONLY in the build container (needs /root/reference, which does not exist
on the GPU box).  The reference is imported, driven with seeded inputs, and its
inputs + outputs are recorded as small .npz fixtures under tests/golden/.
Nothing of the reference's source is stored - only data.

Reference entry points exercised (all under /root/reference/envs):
  TestEnv.my_step            test_env.py:124-266
  TestEnv.my_step_design     test_env.py:269-349
  TestEnv.my_step_ch         test_env.py:351-443
  TestEnv.obtain_state       test_env.py:527-583
  TestEnv.reset_mobility_env test_env.py:479-484
  Network.update_velocity    network.py:208-223
  Network.get_information_age network.py:560-574

  SemiPersistentScheduling   algorithms/v2x_sps.py:8-104   (`sps` fixtures)

Usage:  python tests/golden/gen_golden.py        (rewrites tests/golden/*.npz)
"""
import contextlib
import hashlib
import io
import json
import os
import sys
from unittest import mock

import numpy as np

REF = "/root/reference/envs"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    from test_env import TestEnv  # noqa: E402  (the reference)
    import network as ref_network  # noqa: E402


# The `EnvironmentTest` block of configs/4ue_3r_toy/*_b20_*_dis_07.yaml:45-71,
# restated as data (values only).
TOY = dict(
    congestion_test=True, load_positions=False, num_channels=3, num_users=4,
    mobility=True, mobility_vary=False, highway_length=100,
    enable_fingerprint=False, reward_design=2, communication_range=250,
    State=dict(type=2, add_action=True, add_reward=False, add_index=False,
               add_velocity=False, action_index="binary", piggybacking=False,
               add_position=False, add_positional_dist=False,
               add_positional_dist_piggy=True, add_positional_dist_type=2,
               add_channel_obs=False, num_bins=20),
)


def cfg_with(base=None, state=None, **kw):
    c = json.loads(json.dumps(base or TOY))
    c.update(kw)
    if state:
        c["State"].update(state)
    return c


def big_cfg(N, A, L, **kw):
    return cfg_with(num_users=N, num_channels=A, highway_length=L,
                    congestion_test=False, **kw)


def make_env(cfg):
    with contextlib.redirect_stdout(io.StringIO()):
        return TestEnv(**json.loads(json.dumps(cfg)))


def set_init(env, x0, y0, v0):
    """Overwrite the random topology (network.py:92-119) with recorded values,
    keeping the reference's own scalar types (np.int64 positions, float v)."""
    for u, veh in enumerate(env.network.vehicles):
        veh.pos_x = np.int64(x0[u]) if float(x0[u]).is_integer() else float(x0[u])
        veh.pos_y = np.int64(y0[u]) if float(y0[u]).is_integer() else float(y0[u])
        veh.pos = [veh.pos_x, veh.pos_y]
        veh.velocity = float(v0[u])


def tables(env):
    N = env.NUM_USERS
    seq = np.zeros((N, N), np.int64)
    age = np.zeros((N, N), np.int64)
    tx = np.zeros((N, N), np.float64)
    ty = np.zeros((N, N), np.float64)
    for u, veh in enumerate(env.network.vehicles):
        for k in range(N):
            e = veh.pos_of_neighbors[k]
            seq[u, k] = e["seq_number"]
            age[u, k] = e["last_updated"]
            tx[u, k] = e["xpos"]
            ty[u, k] = e["ypos"]
    return seq, age, tx, ty


def last_arrival(env):
    N = env.NUM_USERS
    la = np.zeros((N, N), np.int64)
    for t in range(N):
        for r in range(N):
            la[t, r] = env.network.last_arrival_time[t][r]
    return la


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(name, cfg, init, steps, vel_updates=None, table_every=1,
             full_tables=True, episode_eps=None, trace=None, trace_after=None, extra=None):
    """steps: list of (mode, actions, t).  vel_updates: {step_index: draws[N]}
    applied AFTER that step (main_test.py:226-233 order)."""
    if trace is not None:
        # trace replay (network.py:171-178): the reference np.load()s `load_file_pos`
        tpath = "/tmp/diral_golden_trace_%s.npy" % name
        np.save(tpath, np.asarray(trace, dtype=np.float64))
        cfg = cfg_with(cfg, load_positions=True, load_file_pos=tpath)
    env = make_env(cfg)
    N, A = env.NUM_USERS, env.NUM_CHANNELS
    if isinstance(init, str) and init == "fixed4":
        env.reset_mobility_env()
    elif isinstance(init, str) and init == "design6":
        pass  # enable_design_topology builds it in the constructor
    else:
        set_init(env, *init)
    x0 = np.array([float(v.pos_x) for v in env.network.vehicles])
    y0 = np.array([float(v.pos_y) for v in env.network.vehicles])
    v0 = np.array([float(v.velocity) for v in env.network.vehicles])

    rec = dict(rews=[], chobs=[], state=[], pos_x=[], vel=[], ia=[])
    tab = dict(step=[], seq=[], age=[], x=[], y=[], la=[])
    tab_sha = []
    vel_updates = vel_updates or {}
    for si, (mode, acts, t) in enumerate(steps):
        acts = np.asarray(acts, dtype=np.int32)
        with contextlib.redirect_stdout(io.StringIO()):
            if mode == "step":
                obs, rews = env.my_step(acts, t)
            elif mode == "ch":
                obs, rews = env.my_step_ch(acts, t)
            elif mode == "design":
                obs, rews = env.my_step_design(acts, t)
            else:
                raise ValueError(mode)
            ep, eps = (episode_eps[si] if episode_eps else (0, 1))
            st = env.obtain_state(obs, acts, list(rews), ep, eps)
        rec["rews"].append(np.array(rews, dtype=np.float64))
        rec["chobs"].append(np.array([obs[u] for u in range(N)], dtype=np.float64))
        rec["state"].append(np.array([np.asarray(s, dtype=np.float64) for s in st]))
        rec["pos_x"].append(np.array([float(v.pos_x) for v in env.network.vehicles]))
        rec["ia"].append(np.array(env.network.get_information_age(t), dtype=np.int64))
        if si in vel_updates:
            draws = list(vel_updates[si])
            with mock.patch.object(ref_network.random, "randrange",
                                   side_effect=lambda a, b: draws.pop(0)):
                env.update_velocity()
        rec["vel"].append(np.array([float(v.velocity) for v in env.network.vehicles]))
        if trace is not None and si == trace_after:
            with contextlib.redirect_stdout(io.StringIO()):
                env.load_saved_positions()                      # main_test.py:118
        seq, age, tx, ty = tables(env)
        la = last_arrival(env)
        tab_sha.append([sha(seq), sha(age), sha(tx), sha(ty), sha(la)])
        if full_tables and (si % table_every == 0 or si == len(steps) - 1):
            tab["step"].append(si)
            tab["seq"].append(seq)
            tab["age"].append(age)
            tab["x"].append(tx)
            tab["y"].append(ty)
            tab["la"].append(la)

    out = dict(
        cfg=np.array(json.dumps(cfg)),
        x0=x0, y0=y0, v0=v0,
        modes=np.array([s[0] for s in steps]),
        actions=np.array([s[1] for s in steps], dtype=np.int32),
        tsteps=np.array([s[2] for s in steps], dtype=np.int64),
        vel_update_steps=np.array(sorted(vel_updates), dtype=np.int64),
        vel_update_draws=np.array([vel_updates[k] for k in sorted(vel_updates)],
                                  dtype=np.uint8).reshape(len(vel_updates), N),
        episode_eps=np.array(episode_eps if episode_eps else
                             [(0, 1)] * len(steps), dtype=np.float64),
        state_space=np.int64(env.get_state_space()),
        table_sha=np.array(tab_sha),
        trace=np.asarray(trace if trace is not None else np.zeros((0, N)), dtype=np.float64),
        trace_after=np.int64(-1 if trace_after is None else trace_after),
    )
    for k, v in rec.items():
        out[k] = np.array(v)
    for k, v in tab.items():
        out["tab_" + k] = np.array(v)
    if getattr(env, "piggybacking", False):
        # TestEnv.prev_obs after the last step (test_env.py:260-261): dict user -> ndarray[A]
        out["prev_obs"] = np.array([env.prev_obs[u] for u in range(N)], dtype=np.float64)
    for k, v in (extra or {}).items():
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s N=%-3d A=%-2d steps=%-3d  %7.1f KB" % (
        name, N, A, len(steps), os.path.getsize(path) / 1024))


def run_driver_case(name, cfg, init, n_prefill, T, enable_channel, global_reward_avg, ia_averaging,
                    episode_interval, seed, ia_penalty_enable=False, ia_penalty_threshold=5, ia_penalty_value=-10):
    """The env-facing call sequence of main_test.marl_test (main_test.py:86-236)
    with the TF agent replaced by recorded random actions: bootstrap, prefill with
    my_step_design (or my_step_ch), slot loop with information age, reward shaping,
    episode boundaries.  utils/misc.calculate_ia_penalty is the reference's own."""
    sys.path.insert(0, "/root/reference")
    from utils.misc import calculate_ia_penalty
    rng = np.random.default_rng(seed)
    env = make_env(cfg)
    N, A = env.NUM_USERS, env.NUM_CHANNELS
    if isinstance(init, str) and init == "fixed4":
        env.reset_mobility_env()
    else:
        set_init(env, *init)
    x0 = np.array([float(v.pos_x) for v in env.network.vehicles])
    y0 = np.array([float(v.pos_y) for v in env.network.vehicles])
    v0 = np.array([float(v.velocity) for v in env.network.vehicles])
    rec = dict(boot_action=None, boot_state=None, pre_actions=[], pre_states=[], actions=[], states=[],
               raw_reward=[], shaped_reward=[], sum_r=[], collision=[], ia=[], ia_sum=[], ia_pen=[],
               episode_end=[], vel_draws=[], episode=[], eps=[])
    with contextlib.redirect_stdout(io.StringIO()):
        action = rng.integers(0, A, size=N).astype(np.int32)           # env.sample() (main_test.py:89)
        obs, rews = env.my_step(action, 0)                               # :92
        rews = list(rews)
        state = env.obtain_state(obs, action, rews)                      # :94
        rec["boot_action"], rec["boot_state"] = action, np.array(state, dtype=np.float64)
        for ii in range(n_prefill):                                      # :99-114
            action = rng.integers(0, A, size=N).astype(np.int32)
            if enable_channel:
                obs, reward = env.my_step_ch(action, 0)
            else:
                obs, reward = env.my_step_design(action, 0)
            next_state = env.obtain_state(obs, action, rews)             # stale `rews` (main_test.py:110)
            rec["pre_actions"].append(action)
            rec["pre_states"].append(np.array(next_state, dtype=np.float64))
        episode, eps, sum_ia_prev = 0, 0.99, 0
        pen_counter = np.zeros(N, int)
        prev_actions = -np.ones(N, int)
        for time_step in range(T):                                       # :119
            action = rng.integers(0, A, size=N).astype(np.int32)
            if enable_channel:
                obs, reward = env.my_step_ch(action, time_step)          # :144
            else:
                obs, reward = env.my_step(action, time_step)             # :146
            raw = np.array(reward, dtype=np.float64)
            ia = env.network.get_information_age(time_step)             # :150
            ia_sum = calculate_ia_penalty(ia)                            # :151
            ia_penalty = 0
            if ia_averaging:                                             # :153-160
                if ia_sum > sum_ia_prev:
                    ia_penalty = -1
                elif ia_sum < sum_ia_prev:
                    ia_penalty = 1
                sum_ia_prev = ia_sum
            next_state = env.obtain_state(obs, action, reward, episode, eps)   # :164
            sum_r = np.sum(reward)                                       # :171
            collision = A - sum_r                                        # :178
            for i in range(len(reward)):                                 # :188-206
                if ia_averaging:
                    reward[i] += ia_penalty
                if ia_penalty_enable:
                    if reward[i] < 1 and action[i] == prev_actions[i]:
                        pen_counter[i] += 1
                    else:
                        pen_counter[i] = 0
                    if pen_counter[i] > ia_penalty_threshold:
                        reward[i] = ia_penalty_value
                    prev_actions[i] = action[i]
                if global_reward_avg:
                    reward[i] = reward[i] + sum_r / len(reward)
            end = (time_step % episode_interval == episode_interval - 1)  # :226
            draws = np.zeros(N, np.uint8)
            if end:
                episode += 1
                eps = max(eps * 0.9992, 0.001)
                draws = rng.integers(1, 4, size=N).astype(np.uint8)
                dl = list(draws)
                with mock.patch.object(ref_network.random, "randrange", side_effect=lambda a, b: dl.pop(0)):
                    env.update_velocity()                                # :233
            for k, v in (("actions", action), ("states", np.array(next_state, dtype=np.float64)),
                         ("raw_reward", raw), ("shaped_reward", np.array(reward, dtype=np.float64)),
                         ("sum_r", sum_r), ("collision", collision), ("ia", np.array(ia)), ("ia_sum", ia_sum),
                         ("ia_pen", ia_penalty), ("episode_end", end), ("vel_draws", draws),
                         ("episode", episode), ("eps", eps)):
                rec[k].append(v)
    seq, age, tx, ty = tables(env)
    out = dict(cfg=np.array(json.dumps(cfg)), x0=x0, y0=y0, v0=v0,
               opts=np.array(json.dumps(dict(n_prefill=n_prefill, T=T, enable_channel=enable_channel,
                                             global_reward_avg=global_reward_avg, ia_averaging=ia_averaging,
                                             episode_interval=episode_interval, ia_penalty_enable=ia_penalty_enable,
                                             ia_penalty_threshold=ia_penalty_threshold,
                                             ia_penalty_value=ia_penalty_value))),
               final_seq=seq, final_age=age, final_x=tx,
               final_pos=np.array([float(v.pos_x) for v in env.network.vehicles]),
               final_vel=np.array([float(v.velocity) for v in env.network.vehicles]))
    for k, v in rec.items():
        out[k] = np.array(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s N=%-3d A=%-2d slots=%-3d  %7.1f KB" % (name, N, A, T, os.path.getsize(path) / 1024))


def rand_init(rng, N, L, vary):
    # network.py:103-110: x=randint(0,L) (integer valued), y=randint(0,1)=0,
    # v = 1.7 if mobility_vary else uniform(1.1, 2.7)
    x0 = rng.integers(0, L, size=N).astype(np.float64)
    y0 = np.zeros(N)
    v0 = np.full(N, 1.7) if vary else rng.uniform(1.1, 2.7, size=N)
    return x0, y0, v0


def rand_steps(rng, mode, T, N, A, sticky=0.0):
    acts = rng.integers(0, A, size=N)
    steps = []
    for t in range(T):
        new = rng.integers(0, A, size=N)
        keep = rng.random(N) < sticky
        acts = np.where(keep, acts, new)
        steps.append((mode, acts.copy(), t))
    return steps


def main():
    toy_actions = [[0, 1, 2, 0], [0, 0, 0, 0], [1, 1, 2, 2], [2, 2, 2, 1]]

    # ---- G1: fixed 4-UE topology, every reward design, both step kinds -----
    for rd in (1, 2, 3, 4, 5):
        run_case("g1_step_rd%d" % rd, cfg_with(reward_design=rd), "fixed4",
                 [("step", a, t) for t, a in enumerate(toy_actions)])
    for rd in (2, 3, 4):
        run_case("g1_ch_rd%d" % rd, cfg_with(reward_design=rd), "fixed4",
                 [("ch", a, t) for t, a in enumerate(toy_actions)])
    run_case("g1_design", cfg_with(), "fixed4",
             [("design", a, 0) for a in toy_actions])
    for nb in (10, 40):
        run_case("g1_step_rd2_b%d" % nb, cfg_with(state=dict(num_bins=nb)),
                 "fixed4", [("step", a, t) for t, a in enumerate(toy_actions)])
    # longer toy run past the ghost-entry phase (19 slots)
    rng = np.random.default_rng(11)
    run_case("g1_step_rd2_long", cfg_with(), "fixed4",
             rand_steps(rng, "step", 60, 4, 3), table_every=10)
    # non-toy weights branch (network.py:291-295) on the toy topology
    run_case("g1_step_rd1_nontoy", cfg_with(reward_design=1, congestion_test=False,
                                            communication_range=1),
             "fixed4", [("step", a, t) for t, a in enumerate(toy_actions)])

    # ---- state-vector flag coverage (test_env.py:49-85, 527-583) -----------
    allflags = dict(add_reward=True, add_index=True, add_velocity=True,
                    add_position=True, add_channel_obs=True)
    run_case("g1_flags_all", cfg_with(state=allflags, enable_fingerprint=True),
             "fixed4", [("step", a, t) for t, a in enumerate(toy_actions)],
             episode_eps=[(0, 1.0), (0, 0.99), (1, 0.98), (1, 0.5)])
    run_case("g1_flags_real_type1",
             cfg_with(state=dict(action_index="real", type=1, add_channel_obs=True)),
             "fixed4", [("step", a, t) for t, a in enumerate(toy_actions)])
    run_case("g1_flags_nopiggy",
             cfg_with(state=dict(add_positional_dist_piggy=False, add_channel_obs=True)),
             "fixed4", [("step", a, t) for t, a in enumerate(toy_actions)])
    run_case("g1_pf", cfg_with(proportional_fair=True), "fixed4",
             [("step", [0, 0, 1, 2], t) for t in range(14)] +
             [("step", [0, 1, 1, 2], 14), ("step", [0, 0, 1, 2], 15)],
             table_every=8)
    # secondary observation modes
    run_case("g1_posdist_full", cfg_with(state=dict(add_positional_dist=True)),
             "fixed4", [("step", a, t) for t, a in enumerate(toy_actions)])
    run_case("g1_posdist_type1", cfg_with(state=dict(add_positional_dist_type=1)),
             "fixed4", [("step", a, t) for t, a in enumerate(toy_actions)])

    # ---- G2: 3-UE line, multi-hop ordering ----------------------
    line = cfg_with(num_users=3, num_channels=3, highway_length=1000,
                    congestion_test=False)
    init3 = (np.array([0., 200., 400.]), np.zeros(3), np.array([1.5, 1.5, 1.5]))
    run_case("g2_line_012", line, init3, [("step", [0, 1, 2], 0), ("step", [0, 1, 2], 1)])
    run_case("g2_line_210", line, init3, [("step", [2, 1, 0], 0), ("step", [2, 1, 0], 1)])

    # ---- G3: 6-UE design topology, two communication ranges ----------------
    for rc in (100, 250):
        d6 = cfg_with(num_users=6, num_channels=4, highway_length=2000,
                      congestion_test=False, enable_design_topology=True,
                      communication_range=rc)
        rng = np.random.default_rng(30 + rc)
        st = rand_steps(rng, "design", 6, 6, 4) + rand_steps(rng, "step", 6, 6, 4) \
            + rand_steps(rng, "ch", 6, 6, 4)
        run_case("g3_design6_rc%d" % rc, d6, "design6", st, table_every=6)

    # ---- G4: C2-shaped 64 UE / 32 res --------------------------------------
    rng = np.random.default_rng(1234)
    c2 = big_cfg(64, 32, 2000)
    run_case("g4_c2_step", c2, rand_init(rng, 64, 2000, False),
             rand_steps(rng, "step", 40, 64, 32), table_every=39, full_tables=True)
    rng = np.random.default_rng(1235)
    run_case("g4_c2_ch", c2, rand_init(rng, 64, 2000, False),
             rand_steps(rng, "ch", 24, 64, 32, sticky=0.5), full_tables=False)
    rng = np.random.default_rng(1236)
    c2v = big_cfg(64, 32, 2000, mobility_vary=True, reward_design=1,
                  state=dict(add_channel_obs=True, add_reward=True))
    run_case("g4_c2_vary_rd1", c2v, rand_init(rng, 64, 2000, True),
             rand_steps(rng, "step", 30, 64, 32, sticky=0.8),
             vel_updates={24: rng.integers(1, 4, size=64)}, full_tables=False)
    rng = np.random.default_rng(1237)
    run_case("g4_c2_design", c2, rand_init(rng, 64, 2000, False),
             rand_steps(rng, "design", 8, 64, 32), full_tables=False)

    # ---- G5: C3-shaped 256 UE / 64 res, congested --------------------------
    rng = np.random.default_rng(2345)
    run_case("g5_c3_step", big_cfg(256, 64, 4000),
             rand_init(rng, 256, 4000, False),
             rand_steps(rng, "step", 5, 256, 64), full_tables=False)

    # ---- G6: C5-shaped 128 UE / 64 res, mobility_vary ----------------------
    rng = np.random.default_rng(3456)
    run_case("g6_c5_vary", big_cfg(128, 64, 4000, mobility_vary=True),
             rand_init(rng, 128, 4000, True),
             rand_steps(rng, "step", 30, 128, 64),
             vel_updates={24: rng.integers(1, 4, size=128)}, full_tables=False)

    # ---- odd sizes: N not a multiple of 64, A > N, A = 1 -------------------
    rng = np.random.default_rng(4567)
    run_case("g8_n70_a5", big_cfg(70, 5, 1500), rand_init(rng, 70, 1500, False),
             rand_steps(rng, "step", 12, 70, 5), full_tables=False)
    rng = np.random.default_rng(4568)
    run_case("g8_n5_a9_ch", big_cfg(5, 9, 300, reward_design=3, communication_range=120),
             rand_init(rng, 5, 300, False),
             rand_steps(rng, "ch", 25, 5, 9), table_every=24)
    rng = np.random.default_rng(4569)
    run_case("g8_n33_a1", big_cfg(33, 1, 800), rand_init(rng, 33, 800, False),
             rand_steps(rng, "step", 4, 33, 1), full_tables=False)


def main_piggyback():
    # ---- P: State.piggybacking (test_env.py:71-79, 241-254, 260-264): my_step returns each agent's observation
    # with the previous observation of every resource's closest transmitter np.insert-ed at the resource's index -
    # A * A values, the channel-observation section of the state vector; defined while every receiver hears a
    # transmitter on every used resource (communication_range >= highway_length), a KeyError otherwise
    toy_actions = [[0, 1, 2, 0], [0, 0, 0, 0], [1, 1, 2, 2], [2, 2, 2, 1]]
    pb = dict(piggybacking=True, add_channel_obs=True)
    rng = np.random.default_rng(501)
    run_case("g1_piggyback", cfg_with(state=pb), "fixed4",
             [("step", a, t) for t, a in enumerate(toy_actions)] +
             [("step", rng.integers(0, 3, size=4), t) for t in range(4, 30)], table_every=29)
    rng = np.random.default_rng(502)
    run_case("g1_piggyback_flags",
             cfg_with(state=dict(pb, add_reward=True, add_index=True, add_position=True, add_velocity=True,
                                 add_positional_dist=True), reward_design=3, enable_fingerprint=True),
             "fixed4", rand_steps(rng, "step", 12, 4, 3, sticky=0.4), table_every=11,
             episode_eps=[(t // 5, 0.99 ** t) for t in range(12)])
    rng = np.random.default_rng(503)
    run_case("g8_piggyback_n6_a4", big_cfg(6, 4, 400, communication_range=500, state=dict(pb, num_bins=10)),
             rand_init(rng, 6, 400, False), rand_steps(rng, "step", 25, 6, 4), table_every=24)
    rng = np.random.default_rng(504)
    run_case("g8_piggyback_n9_a5_type1hist",
             big_cfg(9, 5, 300, communication_range=300, mobility_vary=True,
                     state=dict(pb, add_positional_dist_type=1, num_bins=8)),
             rand_init(rng, 9, 300, True), rand_steps(rng, "step", 30, 9, 5, sticky=0.5),
             vel_updates={24: rng.integers(1, 4, size=9)}, table_every=29)
    # a highway longer than the communication range: the slot in which a receiver hears nobody on a used
    # resource raises KeyError (`self.prev_obs[None]`, test_env.py:243).  Recorded: the slots before it, and the
    # actions of the slot that raised
    rng = np.random.default_rng(505)
    cfg = big_cfg(8, 3, 1000, communication_range=250, state=pb)
    # (the vehicles start within range of one another just short of the end of the highway; the first to wrap
    # around to x = 0 (network.py:189-206) is 900 m from the rest: network.py:318-332 has no ring distance)
    init = (np.arange(8) * 7.0 + 930.0, np.zeros(8), rng.uniform(1.1, 2.7, size=8))
    steps = rand_steps(rng, "step", 40, 8, 3)
    env = make_env(cfg)
    set_init(env, *init)
    raised_at = -1
    for si, (_, acts, t) in enumerate(steps):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                env.my_step(np.asarray(acts, dtype=np.int32), t)
        except KeyError as ex:
            assert ex.args == (None,)
            raised_at = si
            break
    assert raised_at >= 0
    run_case("g8_piggyback_keyerror", cfg, init, steps[:raised_at], table_every=max(raised_at - 1, 1),
             extra=dict(keyerror_actions=np.asarray(steps[raised_at][1], dtype=np.int32),
                        keyerror_t=np.int64(steps[raised_at][2])))


def main_trace():
    # ---- T: trace replay (load_positions, network.py:171-178, 194-199) ----------
    rng = np.random.default_rng(91)
    N, A, L = 12, 5, 500
    tr = np.sort(rng.uniform(0, L, size=(7, N)), axis=1) + rng.normal(0, 3, size=(7, N))
    run_case("g9_trace_replay", big_cfg(N, A, L, communication_range=140), rand_init(rng, N, L, False),
             rand_steps(rng, "step", 6, N, A) + [("step", rng.integers(0, A, size=N), t) for t in (6, 7, 13, 20, 3)]
             + rand_steps(rng, "ch", 4, N, A), trace=tr, trace_after=2, table_every=5)


def main_driver():
    # ---- D: driver-loop fixtures -------------------
    run_driver_case("d1_driver_toy", cfg_with(), "fixed4", n_prefill=12, T=60, enable_channel=False,
                    global_reward_avg=True, ia_averaging=False, episode_interval=25, seed=71)
    rng = np.random.default_rng(72)
    run_driver_case("d2_driver_ch_vary", big_cfg(16, 6, 600, mobility_vary=True, reward_design=3,
                                                 communication_range=150),
                    rand_init(rng, 16, 600, True), n_prefill=8, T=80, enable_channel=True,
                    global_reward_avg=True, ia_averaging=True, episode_interval=25, seed=73,
                    ia_penalty_enable=True, ia_penalty_threshold=2)


def run_sps_case(name, n_agents, A, T, threshold, seed, level_lo, level_hi, tie_step, keep_lo=0.0):
    """The reference's SPS agent (algorithms/v2x_sps.py) itself, one object per agent, fed
    recorded selection windows; its three global-RNG calls are mocked with recorded draws:
      random.randint(a, b)  -> the recorded counter / initial values
      random.random()       -> the recorded keep draw
      random.choice(sB)     -> sB[draw % len(sB)] with the recorded draw
    Recorded: initial (prev_action, counter), per step and agent the window (as integer
    codes: window = code * tie_step, exact in float64), the three draws, and the decisions
    (action, reselection_counter, prev_action after the step) plus how often the threshold
    was raised (len of the while loop, v2x_sps.py:41-50) for the fixture's own statistics."""
    sys.path.insert(0, "/root/reference/algorithms")
    import v2x_sps as ref_sps
    rng = np.random.default_rng(seed)
    cur = {}
    ref_sps.random = mock.MagicMock()

    def fake_randint(a, b):
        if (a, b) == (5, 16):
            return int(cur["counter"])
        raise AssertionError((a, b))
    ref_sps.random.randint.side_effect = fake_randint
    ref_sps.random.random.side_effect = lambda: float(cur["keep"])
    ref_sps.random.choice.side_effect = lambda seq: seq[int(cur["choice"]) % len(seq)]

    init_prev = rng.integers(0, A, size=n_agents)           # randint(0, selection_window) with window = A - 1
    init_cnt = rng.integers(5, 16, size=n_agents)            # randint(5, 15)
    agents = []
    for u in range(n_agents):
        draws = [int(init_prev[u]), int(init_cnt[u])]
        ref_sps.random.randint.side_effect = lambda a, b, d=draws: d.pop(0)
        ag = ref_sps.SemiPersistentScheduling(u, A - 1, threshold)
        assert ag.prev_action == init_prev[u] and ag.reselection_counter == init_cnt[u]
        agents.append(ag)
    ref_sps.random.randint.side_effect = fake_randint

    codes = np.zeros((T, n_agents, A), np.int16)
    d_counter = rng.integers(5, 17, size=(T, n_agents)).astype(np.int32)
    d_keep = keep_lo + (1.0 - keep_lo) * rng.random((T, n_agents))   # keep_lo > 0: more re-selections per recorded step
    d_choice = rng.integers(0, 1 << 20, size=(T, n_agents)).astype(np.int32)
    actions = np.zeros((T, n_agents), np.int32)
    counters = np.zeros((T, n_agents), np.int32)
    prevs = np.zeros((T, n_agents), np.int32)
    for t in range(T):
        for u, ag in enumerate(agents):
            # windows: mostly a busy band around the threshold (ties by quantisation), sometimes
            # everything far above it (several 3 dB raises), sometimes nearly all free
            r = rng.random()
            if r < 0.2:
                c = rng.integers(level_hi, level_hi + 40, size=A)
            elif r < 0.35:
                c = rng.integers(level_lo - 60, level_lo, size=A)
            else:
                c = rng.integers(level_lo, level_hi, size=A)
            codes[t, u] = c
            win = [float(v) * tie_step for v in c]
            cur.update(counter=d_counter[t, u], keep=d_keep[t, u], choice=d_choice[t, u])
            actions[t, u] = ag.step(win)
            counters[t, u] = ag.reselection_counter
            prevs[t, u] = ag.prev_action
    n_resel = ref_sps.random.choice.call_count
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, A=np.int64(A), threshold=np.float64(threshold), tie_step=np.float64(tie_step),
                        init_prev=init_prev.astype(np.int32), init_counter=init_cnt.astype(np.int32),
                        codes=codes, draw_counter=d_counter, draw_keep=d_keep, draw_choice=d_choice,
                        actions=actions, counters=counters, prev_actions=prevs, reselections=np.int64(n_resel))
    print("%-28s agents=%-3d A=%-2d steps=%-3d reselections=%d  %7.1f KB" % (
        name, n_agents, A, T, n_resel, os.path.getsize(path) / 1024))


def main_sps():
    # ---- S: the SPS baseline, recorded from algorithms/v2x_sps.py ----
    # integer-dB windows around an integer threshold: ties, threshold raises
    run_sps_case("s1_sps_int_threshold", n_agents=40, A=12, T=140, threshold=-110.0, seed=81,
                 level_lo=-125, level_hi=-95, tie_step=1.0)
    # non-integer threshold and quarter-dB windows: `tmp_threshold += 3` accumulates roundings
    # ((thr + 3) - 3 != thr), boundary subframes sit within an ulp of the threshold sequence
    run_sps_case("s2_sps_frac_threshold", n_agents=40, A=20, T=140, threshold=-110.3, seed=82,
                 level_lo=-500, level_hi=-380, tie_step=0.25)
    # tiny window (A = 3: min_sA = 0.6) and A = 1 (the only subframe is the previous action:
    # sA stays empty while len(sA) < 0.2 ... the reference loops forever there, so A >= 2)
    run_sps_case("s3_sps_small_window", n_agents=24, A=3, T=200, threshold=-110.0, seed=83,
                 level_lo=-120, level_hi=-100, tie_step=0.5)
    # wide windows: exactly one wavefront of subframes, then 2 and 4 subframes per lane of the
    # wave-cooperative kernel (csrc/aux_kernels.hpp), then beyond it (one thread per agent)
    run_sps_case("s4_sps_window64", n_agents=70, A=64, T=40, threshold=-110.0, seed=84,
                 level_lo=-125, level_hi=-95, tie_step=1.0, keep_lo=0.7)
    run_sps_case("s5_sps_window100", n_agents=30, A=100, T=60, threshold=-110.3, seed=85,
                 level_lo=-460, level_hi=-400, tie_step=0.25, keep_lo=0.7)
    run_sps_case("s6_sps_window200", n_agents=20, A=200, T=60, threshold=-110.0, seed=86,
                 level_lo=-118, level_hi=-102, tie_step=1.0, keep_lo=0.7)
    run_sps_case("s7_sps_window300", n_agents=12, A=300, T=70, threshold=-110.0, seed=87,
                 level_lo=-240, level_hi=-200, tie_step=0.5, keep_lo=0.7)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "driver":
        main_driver()
    elif len(sys.argv) > 1 and sys.argv[1] == "sps":
        main_sps()
    elif len(sys.argv) > 1 and sys.argv[1] == "trace":
        main_trace()
    elif len(sys.argv) > 1 and sys.argv[1] == "piggyback":
        main_piggyback()
    else:
        main()
        main_piggyback()
        main_trace()
        main_driver()
        main_sps()
