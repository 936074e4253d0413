"""A K-slot rollout captured into ONE hipGraph (launch-overhead-free closed loop).

The reference's slot loop is env step -> reward shaping -> policy, once per slot (main_test.py:119-236); on the
device that is three launches per slot driven from Python (`examples/rollout_sps.py`: ~100 us per slot at 4096 envs
against ~90 us of kernels).  `GraphRollout` records K slots of

    obs, rew = env.my_step*(actions, t)            (state + reward + channel observation, one launch)
    shaped   = diral_driver_shape(rew, actions)    (main_test.py:171-206)
    actions  = SpsPolicy(obs, actions)             (algorithms/v2x_sps.py, decision from the channel observation)

into one graph and replays it.  Everything a replay must see move on lives in device memory: the slot number
(`diral_env_set_clock`: `done`, arrival stamps, trace replay), the seed offset of the policy's draws
(`diral_sps_step_chobs_clocked`), the actions (two buffers the slots alternate between), and the clock itself,
advanced by K by the graph's last node.  A replay of the graph equals K eager slots bit for bit
(tests/test_gpu_parity.py::test_graph_rollout_equals_eager).

Scope: the per-slot outputs live in the env's output ring (`io_ring` sets; K must be a multiple of it), i.e. a
replay leaves the LAST `io_ring` slots' state / reward behind - a policy-evaluation rollout (metrics accumulate on
the device), not a replay-buffer filler.  Episode ends that change the env from the host (`update_velocity` with
mobility_vary, the epsilon schedule behind the fingerprint columns) are not part of the graph: such configs raise.
"""
from __future__ import annotations

import contextlib
import ctypes
import gc
from typing import Optional

import torch

from .config import STEP_MY_STEP, STEP_MY_STEP_CH
from .sps import SpsPolicy
from .vec_env import DiralError, VecV2VEnv


@contextlib.contextmanager
def no_finalizers_during_capture():
    """No finalizer may run inside a stream capture: a dead handle's `diral_env_destroy` (or a dead tensor's block going
    back to the driver) is a hipFree, which a capture in the default (global) mode turns into a fatal error of the whole
    process.  Collects what is dead NOW and holds the cyclic collector until the capture has ended."""
    was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was_on:
            gc.enable()


class SlotClock:
    """An int64 slot counter in HBM shared by the env and the policy of a captured rollout."""

    def __init__(self, device, start: int = 0):
        self.t = torch.full((1,), int(start), dtype=torch.int64, device=device)

    def ptr(self) -> int:
        return self.t.data_ptr()

    def value(self) -> int:
        return int(self.t.item())


def shape_rewards(env: VecV2VEnv, reward: torch.Tensor, actions: torch.Tensor, out: torch.Tensor, sum_r: torch.Tensor,
                  coll: torch.Tensor, global_reward_avg: bool) -> None:
    """main_test.py:171-206 without the information-age terms: `diral_driver_shape`, one launch, caller-owned outputs."""
    B, N = reward.shape
    st = env.lib.diral_driver_shape(B, N, env.A, reward.data_ptr(), 1 if reward.dtype == torch.float64 else 0,
                                    actions.data_ptr(), None, None, None, None, 1 if global_reward_avg else 0, 0, 0.0,
                                    out.data_ptr(), sum_r.data_ptr(), coll.data_ptr(), None, None, env._stream())
    if st != 0:
        raise DiralError(st, "diral_driver_shape")


class GraphRollout:
    def __init__(self, env: VecV2VEnv, policy: SpsPolicy, K: int, global_reward_avg: bool = True, enable_channel: bool = False,
                 clock: Optional[SlotClock] = None, capture: bool = True, fused: bool = False):
        cfg = env.cfg
        if cfg.mobility_vary or cfg.enable_fingerprint:
            raise ValueError("GraphRollout: configs whose episode ends act on the env from the host (mobility_vary, "
                             "enable_fingerprint) are not captured")
        if K < 1 or K % env.io_ring or K % 2:
            raise ValueError("K must be a positive multiple of 2 and of the env's io_ring (%d)" % env.io_ring)
        self.env, self.pol, self.K = env, policy, int(K)
        self.fused = bool(fused)       # the slot as ONE launch (diral_env_step_policy) instead of three
        self.global_reward_avg = bool(global_reward_avg)
        self.mode = STEP_MY_STEP_CH if enable_channel else STEP_MY_STEP
        dev = env.device
        self.clock = clock or SlotClock(dev)
        B, N = env.B, env.N
        self.actions = [policy.prev_action.clone(), torch.empty_like(policy.prev_action)]
        dt = env.out_dtype
        self.shaped = [torch.empty((B, N), dtype=dt, device=dev) for _ in range(2)]
        self.sum_r = [torch.empty((B,), dtype=dt, device=dev) for _ in range(2)]
        self.coll = [torch.empty((B,), dtype=dt, device=dev) for _ in range(2)]
        self.graph = None
        self._phase = None
        self._slots_run = 0
        self._closed = False
        env.set_clock(self.clock.t)            # the env holds the tensor from here on (VecV2VEnv.set_clock)
        try:
            if capture:
                # a first eager pass: output buffers exist, the ring <-> plane state is settled (a conversion launch
                # must not be captured: DIRAL_ERR_CAPTURE), lazy initialisations are done
                self._run_slots(eager=True)
                torch.cuda.synchronize(dev)
                self.graph = torch.cuda.CUDAGraph()
                self._stream = torch.cuda.Stream(device=dev)
                self._stream.wait_stream(torch.cuda.current_stream(dev))
                # K a multiple of 3: the captured launches rotate the slow-env sets like eager ones (the list of the
                # envs to dispatch first stays live from replay to replay instead of ageing: VecV2VEnv.set_capture_rotation;
                # run() keeps the phase aligned)
                self._phase = env.set_capture_rotation(True) if self.K % 3 == 0 else None
                try:
                    with no_finalizers_during_capture(), torch.cuda.stream(self._stream):
                        with torch.cuda.graph(self.graph, stream=self._stream):
                            self._run_slots(eager=False)
                finally:
                    if self._phase is not None:
                        env.set_capture_rotation(False)
                torch.cuda.current_stream(dev).wait_stream(self._stream)
        except BaseException:
            # (DIRAL_ERR_CAPTURE, an unsupported reward design for my_step_ch, ...): the env must not keep
            # reading a clock that belongs to a rollout that never came to be
            self.graph = None
            self.close()
            raise

    def _one_slot(self, k: int) -> None:
        env, pol = self.env, self.pol
        a, a_next = self.actions[k & 1], self.actions[(k + 1) & 1]
        i = k & 1
        if self.fused:
            env.step_policy(a, k, pol, a_next, shaped_out=self.shaped[i], sum_r_out=self.sum_r[i], collision_out=self.coll[i],
                            global_reward_avg=self.global_reward_avg, clock=self.clock, seed_offset=k, mode=self.mode)
            return
        env._step(self.mode, a, k, want_chobs=True)                      # slot number = clock + k, on the device
        shape_rewards(env, env._rew, a, self.shaped[i], self.sum_r[i], self.coll[i], self.global_reward_avg)
        pol.step_from_chobs_clocked(env._chobs, a, self.clock, k, out=a_next)

    def _run_slots(self, eager: bool) -> None:
        for k in range(self.K):
            self._one_slot(k)
        st = self.env.lib.diral_clock_add(ctypes.c_void_p(self.clock.ptr()), self.K, self.env._stream())
        if st != 0:
            raise DiralError(st, "diral_clock_add")
        if eager:
            self._slots_run += self.K

    def run(self, replays: int = 1) -> None:
        """`replays` x K slots (enqueued on the current stream; no host sync)."""
        if self.graph is not None and getattr(self, "_phase", None) is not None and replays > 0:
            self.env.align_phase(self._phase)        # (eager steps since the last replay may have moved it)
        for _ in range(replays):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._run_slots(eager=True)
                continue
            self._slots_run += self.K

    @property
    def slots(self) -> int:
        return self._slots_run

    def last(self):
        """(state, shaped reward, actions that produced them) of the last slot run."""
        i = (self.K - 1) & 1
        return self.env._obs, self.shaped[i], self.actions[i]

    def close(self) -> None:
        """Remove the slot clock from the env (idempotent).  Also runs on `with`-exit and when the object is dropped."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        env = self.env
        if getattr(env, "_h", None) and env._clock is self.clock.t:
            env.set_clock(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
