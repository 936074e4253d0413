"""The env-facing half of the reference driver, vectorised.

``main_test.marl_test`` (main_test.py:86-236) interleaves the TF agent with a
fixed sequence of env calls and some reward post-processing.  This module keeps
that sequence - bootstrap, prefill, slot loop, information age, reward shaping,
episode boundary - for B envs at once on top of :class:`VecV2VEnv` (or any
object with its method names), so a PyTorch agent can be dropped where the
reference's ``mainDRQN`` sat.  The agent itself is out of scope.

    loop = DriverLoop(env, global_reward_avg=True, episode_interval=25)
    state = loop.bootstrap()                         # main_test.py:89-94
    for _ in range(prefill): state = loop.prefill_step(env.sample())     # :99-114
    for t in range(time_slots):
        out = loop.slot(policy(state), t)            # :119-236, one fused launch + small torch ops
        state = out["next_state"]
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch


def calculate_ia_penalty(ia: torch.Tensor) -> torch.Tensor:
    """utils/misc.py:1-12: sum over bins of (i+1)*ia[i] (bins with ia > 0);
    ia [..., 100] integer -> [...] int64."""
    ia = torch.as_tensor(ia).to(torch.int64)
    w = torch.arange(1, ia.shape[-1] + 1, dtype=torch.int64, device=ia.device)
    return (torch.clamp(ia, min=0) * w).sum(-1)


def np_sum_lastdim(a: torch.Tensor) -> torch.Tensor:
    """Sum over the last axis in NumPy's order (`np.sum(reward)`, main_test.py:171):
    numpy/_core/src/umath/loops_utils.h pairwise_sum - fewer than 8 elements
    sequentially, otherwise eight running accumulators combined as
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail, blocks above 128
    elements split recursively.  Float addition is not associative; keeping the
    order keeps the driver's shaped rewards bit-identical."""
    n = a.shape[-1]
    if n < 8:
        res = torch.zeros_like(a[..., 0])
        for i in range(n):
            res = res + a[..., i]
        return res
    if n <= 128:
        r = [a[..., j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = r[j] + a[..., i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[..., i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return np_sum_lastdim(a[..., :n2]) + np_sum_lastdim(a[..., n2:])


class DriverLoop:
    """Call-sequence and reward post-processing of main_test.py:86-236."""

    def __init__(self, env: Any, enable_channel: bool = False, global_reward_avg: bool = False,
                 ia_averaging: bool = False, ia_penalty_enable: bool = False, ia_penalty_threshold: int = 5,
                 ia_penalty_value: float = -10, episode_interval: int = 25,
                 eps_init: float = 0.99, eps_decay: float = 0.9992, eps_min: float = 0.001,
                 fused: bool = False, device_shaping: Optional[bool] = None):
        self.env = env
        self.enable_channel = enable_channel                  # main_test.py:40
        self.global_reward_avg = global_reward_avg            # :38
        self.ia_averaging = ia_averaging                      # :43
        self.ia_penalty_enable = ia_penalty_enable            # :42
        self.ia_penalty_threshold = ia_penalty_threshold      # :50
        self.ia_penalty_value = ia_penalty_value              # :51
        self.episode_interval = episode_interval              # :31
        self.eps, self.eps_decay, self.eps_min = eps_init, eps_decay, eps_min   # algorithms/policies.py:38-77
        # fused=True: my_step + obtain_state in ONE launch (VecV2VEnv.step semantics; same
        # values, but the channel observation `obs` is not materialised)
        self.fused = fused
        # device_shaping: the reward post-processing of a slot as ONE launch of `diral_driver_shape`
        # (include/diral_env.h) instead of ~70 small torch ops; None = whenever the env is the HIP one.
        # The torch statement below stays: it is what runs on the CPU (oracle-backed tests) and what the
        # kernel is tested against.
        self.device_shaping = (hasattr(env, "lib") and hasattr(env, "_stream")) if device_shaping is None else device_shaping
        # the env rotates >= 2 output-buffer sets (VecV2VEnv(io_ring=2)): what a slot returns stays intact
        # during the next slot, no defensive copies
        self._own_outputs = getattr(env, "io_ring", 1) < 2
        self.episode = 0
        self.N = env.get_total_users()
        self.A = env.get_action_space()
        self._rews0: Optional[torch.Tensor] = None
        self._sum_ia_prev: Optional[torch.Tensor] = None
        self._pen_counter: Optional[torch.Tensor] = None
        self._prev_actions: Optional[torch.Tensor] = None

    @staticmethod
    def _t(x) -> torch.Tensor:
        return x if isinstance(x, torch.Tensor) else torch.as_tensor(x)

    def _actions(self, a) -> torch.Tensor:
        a = self._t(a)
        return a if a.dim() == 2 else a.unsqueeze(0)

    # main_test.py:89-94
    def bootstrap(self, action=None) -> torch.Tensor:
        action = self.env.sample() if action is None else action
        obs, rews = self.env.my_step(action, 0)
        state = self._t(self.env.obtain_state(obs, action, rews)).clone()
        self._rews0 = self._t(rews).clone()
        return state

    # main_test.py:99-114 (the stored reward is the stale bootstrap `rews`, :110-112)
    def prefill_step(self, action) -> torch.Tensor:
        if self.enable_channel:
            obs, _ = self.env.my_step_ch(action, 0)
        else:
            obs, _ = self.env.my_step_design(action, 0)
        return self._t(self.env.obtain_state(obs, action, self._rews0)).clone()

    # main_test.py:99-114 as a whole: K random slots, every state kept (what the loop hands to memory.add)
    def prefill(self, slots: int, seed: int):
        """``for k in range(slots): a = env.sample(seed + k); states[k] = prefill_step(a)``: returns
        ``(states [K, B, N, S], actions [K, B, N])``.  On the HIP env with `enable_channel` off this is ONE launch
        (`VecV2VEnv.prefill` -> `diral_env_prefill`: the env stays on the chip for the K slots); configurations that
        launch does not take - and any other env - run the loop."""
        env = self.env
        if not self.enable_channel and hasattr(env, "prefill"):
            from .config import ERR_UNSUPPORTED
            from .vec_env import DiralError
            try:
                states, acts, _ = env.prefill(env.sample(seed), slots, seed, rew_in=self._rews0)
                return states, acts
            except DiralError as exc:
                if exc.status != ERR_UNSUPPORTED:
                    raise
        states, acts = [], []
        for k in range(int(slots)):
            a = env.sample(seed + k)
            acts.append(self._t(a).clone())
            states.append(self.prefill_step(a))
        return torch.stack(states), torch.stack(acts)

    # main_test.py:119-236, everything between the agent's action and memory.add
    def slot(self, action, time_step: int, want_ia: Optional[bool] = None) -> Dict[str, Any]:
        env = self.env
        fused_state = None
        if self.fused:
            from .config import STEP_MY_STEP, STEP_MY_STEP_CH
            mode = STEP_MY_STEP_CH if self.enable_channel else STEP_MY_STEP
            fused_state, reward, _ = env._step(mode, env._actions(action), time_step, self.episode, self.eps)
            obs = None
        elif self.enable_channel:
            obs, reward = env.my_step_ch(action, time_step)                  # :144
        else:
            obs, reward = env.my_step(action, time_step)                     # :146
        reward_ret = reward                      # as returned: what main_test.py:164 hands to obtain_state
        reward = self._t(reward)
        if self._own_outputs:
            reward = reward.clone()
        raw = reward.clone() if self._own_outputs else reward    # (nothing below writes `reward` in place)
        out: Dict[str, Any] = {}
        need_ia = self.ia_averaging if want_ia is None else want_ia
        ia_penalty = None
        if need_ia:
            ia = self._t(env.info_age(time_step))                            # :150
            out["ia"] = ia
        if need_ia and not self.device_shaping:
            ia_sum = calculate_ia_penalty(ia)                                # :151
            out["ia_sum"] = ia_sum
            if self.ia_averaging:                                            # :153-160
                prev = self._sum_ia_prev if self._sum_ia_prev is not None else torch.zeros_like(ia_sum)
                ia_penalty = torch.where(ia_sum > prev, -1, torch.where(ia_sum < prev, 1, 0)).to(reward.dtype)
                self._sum_ia_prev = ia_sum
                out["ia_penalty"] = ia_penalty
        if fused_state is not None:
            next_state = fused_state.clone() if self._own_outputs else fused_state
        else:
            # (the returned tensors, unmodified: VecV2VEnv then serves the state its fused launch
            # already built instead of a second launch)
            next_state = self._t(env.obtain_state(obs, action, reward_ret, self.episode, self.eps))   # :164
            if self._own_outputs:
                next_state = next_state.clone()
        a = self._actions(action).to(reward.device)
        if self.device_shaping:
            shaped, sum_r, collision, pen, ia_sum = self._shape_on_device(reward, a, out.get("ia"))
            if ia_sum is not None:
                out["ia_sum"] = ia_sum
            if pen is not None:
                out["ia_penalty"] = pen
            episode_end = (time_step % self.episode_interval) == self.episode_interval - 1   # :226
            out.update(next_state=next_state, reward=shaped, raw_reward=raw, sum_r=sum_r, collision=collision,
                       episode_end=episode_end, episode=self.episode, eps=self.eps)
            return out
        sum_r = np_sum_lastdim(reward)                                       # :171 (NumPy's summation order)
        collision = self.A - sum_r                                           # :178
        if ia_penalty is not None:
            reward = reward + ia_penalty.unsqueeze(-1)                       # :190-192
        if self.ia_penalty_enable:                                           # :194-203
            if self._pen_counter is None:
                self._pen_counter = torch.zeros_like(a, dtype=torch.int64)
                self._prev_actions = torch.full_like(a, -1)
            stuck = (reward < 1) & (a == self._prev_actions)
            self._pen_counter = torch.where(stuck, self._pen_counter + 1, torch.zeros_like(self._pen_counter))
            reward = torch.where(self._pen_counter > self.ia_penalty_threshold,
                                 torch.as_tensor(float(self.ia_penalty_value), dtype=reward.dtype, device=reward.device),
                                 reward)
            self._prev_actions = a.clone()
        if self.global_reward_avg:
            # (tensor / tensor: a true IEEE division like Python's `sum_r / len(reward)`; tensor / Python scalar
            # is computed on the GPU as a multiplication by the rounded reciprocal - one ulp off unless N is a
            # power of two)
            n_t = torch.as_tensor(float(self.N), dtype=reward.dtype, device=reward.device)
            reward = reward + (sum_r / n_t).unsqueeze(-1)                    # :205-206
        episode_end = (time_step % self.episode_interval) == self.episode_interval - 1   # :226
        out.update(next_state=next_state, reward=reward, raw_reward=raw, sum_r=sum_r, collision=collision,
                   episode_end=episode_end, episode=self.episode, eps=self.eps)
        return out

    def _shape_on_device(self, reward: torch.Tensor, a: torch.Tensor, ia: Optional[torch.Tensor]):
        """main_test.py:171-206 through `diral_driver_shape` (one launch)."""
        env = self.env
        B, N = reward.shape
        dev = reward.device
        reward = reward.contiguous()
        a32 = a.to(torch.int32).contiguous()
        if self.ia_averaging and self._sum_ia_prev is None:
            self._sum_ia_prev = torch.zeros((B,), dtype=torch.int64, device=dev)
        if self.ia_penalty_enable and self._pen_counter is None:
            self._pen_counter = torch.zeros((B, N), dtype=torch.int32, device=dev)
            self._prev_actions = torch.full((B, N), -1, dtype=torch.int32, device=dev)
        shaped = torch.empty_like(reward)
        sum_r = torch.empty((B,), dtype=reward.dtype, device=dev)
        coll = torch.empty((B,), dtype=reward.dtype, device=dev)
        pen = torch.zeros((B,), dtype=torch.int32, device=dev) if (self.ia_averaging and ia is not None) else None
        ia32 = None if ia is None else ia.to(torch.int32).contiguous()
        ia_sum = None if ia is None else torch.empty((B,), dtype=torch.int64, device=dev)
        # (the information-age term only when this slot fetched the histogram: slot(..., want_ia=False) with
        # ia_averaging skips it, exactly like the torch statement in slot())
        use_ia = self.ia_averaging and ia is not None
        flags = (1 if self.global_reward_avg else 0) | (2 if use_ia else 0) | (4 if self.ia_penalty_enable else 0)

        def p(t):
            return None if t is None else t.data_ptr()
        st = env.lib.diral_driver_shape(B, N, self.A, p(reward), 1 if reward.dtype == torch.float64 else 0, p(a32), p(ia32),
                                        p(self._sum_ia_prev) if use_ia else None,
                                        p(self._pen_counter) if self.ia_penalty_enable else None,
                                        p(self._prev_actions) if self.ia_penalty_enable else None, flags,
                                        int(self.ia_penalty_threshold), float(self.ia_penalty_value), p(shaped), p(sum_r),
                                        p(coll), p(ia_sum), p(pen), env._stream())
        if st != 0:
            raise RuntimeError("diral_driver_shape failed with status %d" % st)
        self._keep = (reward, a32, ia32)
        return shaped, sum_r, coll, (pen.to(reward.dtype) if pen is not None else None), ia_sum

    # main_test.py:226-233: call when out["episode_end"]; `draws` are the
    # random.randrange(1,4) values of Network.update_velocity (None = device RNG)
    def end_episode(self, draws=None) -> None:
        self.episode += 1
        self.eps = max(self.eps * self.eps_decay, self.eps_min)
        if draws is None:
            self.env.update_velocity()
        else:
            self.env.update_velocity(draws)
