"""Drop-in ``TestEnv``: the reference's single-env Python surface on the GPU path.

``algorithms/drl_drqn.py`` / ``ps_drqn.py`` / ``ps_dqn.py`` / ``ps_ppo.py`` touch
the env only through ``get_total_users/get_state_space/get_action_space``
(drl_drqn.py:30,38,39) and ``main_test.py`` drives it with ``sample``,
``my_step*``, ``obtain_state``, ``update_velocity``, ``get_x_pos``,
``reset_mobility_env`` and ``env.network.get_information_age`` (main_test.py:89-233).
This class keeps those names, argument meaning and RETURN SHAPES:

    obs, rews = env.my_step(actions, t)      # obs: dict user -> ndarray[A] (A * A with State.piggybacking); rews: ndarray[N]
    state = env.obtain_state(obs, actions, rews, episode, eps)   # list of N float64 vectors

on top of a B=1 :class:`VecV2VEnv` (float64 outputs).  Every call synchronises
and copies to the host - it exists for compatibility, not speed; training code
that wants throughput uses :class:`VecV2VEnv` directly.
"""
from __future__ import annotations

from typing import Any, Dict, List, Mapping, Optional, Sequence, Union

import numpy as np

from .config import ConfigError, EnvConfig


class _NetworkView:
    """``env.network`` reach-through used by the driver (main_test.py:150)."""

    def __init__(self, owner: "TestEnv"):
        self._o = owner

    def get_information_age(self, timestep: int) -> List[int]:           # network.py:560-574
        return [int(v) for v in self._o._to_np(self._o._env.info_age(timestep))[0]]

    def get_x_positions(self) -> List[float]:                            # network.py:160-169
        return self._o.get_x_pos()

    def get_velocity(self, user: int) -> float:                          # network.py:400-401
        return float(self._o._to_np(self._o._env.export_state(tables=False)["vel"])[0, user])

    def update_velocity(self) -> None:                                   # network.py:208-223
        self._o._env.update_velocity()


class TestEnv:
    """``TestEnv(**config["EnvironmentTest"])`` (envs/test_env.py:6-107)."""

    __test__ = False   # not a pytest class

    def __init__(self, backend: Any = None, device: str = "cuda:0", **kwargs: Any):
        if not kwargs:
            raise ConfigError("TestEnv needs the EnvironmentTest keys (test_env.py:12-48)")
        cfg = EnvConfig.from_dict(kwargs, track_arrival=True)
        self.cfg = cfg
        self.NUM_USERS = cfg.num_users
        self.NUM_CHANNELS = cfg.num_channels
        self.state_space = cfg.state_space
        self.action_space = cfg.num_channels
        self.mobility_vary = cfg.mobility_vary
        if backend is None:
            import torch
            from .vec_env import VecV2VEnv
            backend = VecV2VEnv(cfg, batch=1, device=device, out_dtype=torch.float64)
        self._env = backend
        # the reference constructor draws a random topology (network.py:92-119)
        # or builds the design topology (network.py:69-79)
        if cfg.enable_design_topology:
            self._env.reset_design_topology()
        else:
            self._env.reset_topology(seed=int(np.random.randint(0, 2**31 - 1)))
        self.network = _NetworkView(self)

    # ---- helpers ------------------------------------------------------------
    @staticmethod
    def _to_np(t) -> np.ndarray:
        return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)

    def _obs_dict(self, chobs) -> Dict[int, np.ndarray]:
        c = self._to_np(chobs)[0].astype(np.float64)
        return {u: c[u].copy() for u in range(self.NUM_USERS)}

    # ---- reference surface ------------------------------------------------
    def sample(self) -> np.ndarray:                                       # test_env.py:116-122
        return self._to_np(self._env.sample())[0].astype(np.int64)

    def my_step(self, actions: Sequence[int], timestep: int):             # test_env.py:124-266
        chobs, rews = self._env.my_step(np.asarray(actions, dtype=np.int32), timestep)
        if self.cfg.State.piggybacking:
            # a receiver that hears nobody on a used resource: the reference's `self.prev_obs[tx_id]` with
            # tx_id None raises KeyError inside my_step (test_env.py:243)
            from .config import ERR_PIGGY_NO_TX
            try:
                self._env.check()
            except RuntimeError as exc:
                if getattr(exc, "status", 0) == ERR_PIGGY_NO_TX:
                    raise KeyError(None) from exc
                raise
        return self._obs_dict(chobs), self._to_np(rews)[0].astype(np.float64).copy()

    def my_step_ch(self, actions: Sequence[int], time_step: int):         # test_env.py:351-443
        chobs, rews = self._env.my_step_ch(np.asarray(actions, dtype=np.int32), time_step)
        return self._obs_dict(chobs), self._to_np(rews)[0].astype(np.float64).copy()

    def my_step_design(self, actions: Sequence[int], timestep: int):      # test_env.py:269-349
        chobs, rews = self._env.my_step_design(np.asarray(actions, dtype=np.int32), timestep)
        return self._obs_dict(chobs), self._to_np(rews)[0].astype(np.float64).copy()

    def obtain_state(self, obs: Union[Mapping[int, np.ndarray], np.ndarray], acts: Sequence[int],
                     rewards: Sequence[float], episode_number: float = 0, epsilon: float = 1
                     ) -> List[np.ndarray]:                               # test_env.py:527-583
        if isinstance(obs, Mapping):
            chobs = np.stack([np.asarray(obs[u], dtype=np.float64) for u in range(self.NUM_USERS)])
        else:
            chobs = np.asarray(obs, dtype=np.float64)
        st = self._env.obtain_state(chobs[None], np.asarray(acts, dtype=np.int32),
                                    np.asarray(rewards, dtype=np.float64)[None], episode_number, epsilon)
        st = self._to_np(st)[0].astype(np.float64)
        return [st[u].copy() for u in range(self.NUM_USERS)]

    def get_x_pos(self) -> List[float]:                                   # test_env.py:471-476
        return [float(v) for v in self._to_np(self._env.get_x_pos())[0]]

    def reset_mobility_env(self) -> None:                                 # test_env.py:479-484
        self._env.reset_mobility_env()

    def get_total_users(self) -> int:                                     # test_env.py:486
        return self.NUM_USERS

    def get_num_ch(self) -> int:                                          # test_env.py:489
        return self.NUM_CHANNELS

    def get_state_space(self) -> int:                                     # test_env.py:492
        return self.state_space

    def get_action_space(self) -> int:                                    # test_env.py:495
        return self.action_space

    def update_velocity(self, draws: Optional[Sequence[int]] = None) -> None:   # test_env.py:498-504
        if self.mobility_vary:
            self._env.update_velocity(draws)

    def load_saved_positions(self) -> None:                               # test_env.py:109-114
        if self.cfg.load_positions:
            print("Load the saved positions !!!")
            self._env.load_saved_positions(self.cfg.extra.get("load_file_pos"))
        else:
            print("Load the saved positions disabled !!!")

    def one_hot(self, num: int, len: int) -> np.ndarray:                  # test_env.py:585-595
        assert num >= 0 and num < len, "error"
        vec = np.zeros([len], np.int32)
        vec[num] = 1
        return vec
