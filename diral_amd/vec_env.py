"""VecV2VEnv - B parallel DIRAL environments behind the reference's env surface.

The reference env (envs/test_env.py:6 ``TestEnv``) is one Python object per
process.  This class keeps its method names and argument meaning, batched over
B independent envs that live on one MI355X and advance in ONE fused HIP launch
per time-slot (csrc/step_kernel.hpp) through the C-ABI of include/diral_env.h.

PyTorch is plumbing only: it owns device memory for the I/O tensors and supplies
the current HIP stream.  All arithmetic happens inside libdiral_env.so; there
is no eager/torch fallback - a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from .config import (DT_F32, DT_F64, ERR_HIP, M_COLUMNS, OK, OPT_ENV_OFFSET, OPT_KERNEL_PATH, PATH_AUTO,
                     PATH_GENERAL, PATH_LARGE, STEP_DESIGN, STEP_MY_STEP, STEP_MY_STEP_CH, ConfigError, EnvConfig)

# MA_NeighborTableEntry (envs/ma_messages_pb2.py:195-230) as a host view of DiralNeighborEntry
ENTRY_DTYPE = np.dtype([("pos_x", "<f4"), ("pos_y", "<f4"), ("seq_num", "<i4"), ("last_update", "<i4")])

_MODES = {"my_step": STEP_MY_STEP, "my_step_ch": STEP_MY_STEP_CH, "my_step_design": STEP_DESIGN,
          STEP_MY_STEP: STEP_MY_STEP, STEP_MY_STEP_CH: STEP_MY_STEP_CH, STEP_DESIGN: STEP_DESIGN}


_PARKED: list = []          # (lib, handle) pairs whose owners died inside a stream capture (VecV2VEnv.close)


def _capturing() -> bool:
    try:
        return bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
    except Exception:
        return False


class DiralError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        msg = "%s: %s (status %d)" % (where, _lib.strerror(status), status)
        if detail:
            msg += " - " + detail
        super().__init__(msg)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class VecV2VEnv:
    """B independent V2V resource-allocation envs on one GPU.

    Parameters
    ----------
    cfg : EnvConfig | dict
        the reference's ``EnvironmentTest`` block (dict accepted verbatim).
    batch : int
        number of parallel envs B.
    device : torch.device | str | int
        a CUDA(HIP) device; CPU is rejected.
    out_dtype : torch.float32 | torch.float64
        element type of obs / reward / channel-obs outputs.  Arithmetic is
        always float64 (like the reference); float32 is a final cast.
    step_mode : "my_step" | "my_step_ch" | "my_step_design"
        which reference step function ``step()`` stands for.
    env_offset : int
        global index of this handle's env 0 when a batch is sharded over several
        handles / GPUs: device RNG draws are a function of (seed, global env index),
        so shards reproduce the unsharded batch (``DIRAL_OPT_ENV_OFFSET``).
    speculate_state : bool
        the reference's per-slot pair ``obs, rews = env.my_step(a, t)`` /
        ``env.obtain_state(obs, a, rews, ...)`` (main_test.py:144-164) as ONE fused
        launch: ``my_step*`` also builds the state vector, and ``obtain_state``
        returns it when it is called with exactly what ``my_step*`` returned and
        nothing changed the env in between (otherwise it launches
        ``diral_env_observe`` as before).
    io_ring : int
        number of output-buffer sets (state, reward, done, channel observation) the step
        calls rotate through.  1 (default): the returned tensors are overwritten by the next
        step.  K > 1: they stay intact for K - 1 further steps - a slot loop that needs
        ``state`` and ``next_state`` side by side (main_test.py:207-214, memory.add) takes
        ``io_ring=2`` instead of cloning 13 KB per env and slot.
    """

    def __init__(self, cfg: Union[EnvConfig, Mapping[str, Any]], batch: int = 1,
                 device: Union[str, int, torch.device] = "cuda:0",
                 out_dtype: torch.dtype = torch.float32, step_mode: Union[str, int] = "my_step",
                 env_offset: int = 0, speculate_state: bool = True, io_ring: int = 1,
                 out_buffers: Optional[Mapping[str, Optional[torch.Tensor]]] = None):
        if not isinstance(cfg, EnvConfig):
            cfg = EnvConfig.from_dict(cfg)
        cfg.validate()
        if out_dtype not in (torch.float32, torch.float64):
            raise ValueError("out_dtype must be float32 or float64")
        if step_mode not in _MODES:
            raise ValueError("unknown step_mode %r" % (step_mode,))
        self.cfg = cfg
        self.device = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
        if self.device.type != "cuda":
            raise DiralError(-5, "VecV2VEnv", "a HIP device is required; there is no CPU path")
        if not torch.cuda.is_available():
            raise DiralError(-5, "VecV2VEnv", "torch sees no GPU")
        self.lib = _lib.load()
        self.B = int(batch)
        self.N = cfg.num_users
        self.A = cfg.num_channels
        self.S = cfg.state_space
        self.CW = cfg.chobs_width            # A, or A * A with State.piggybacking (test_env.py:263-264)
        self.out_dtype = out_dtype
        self._dt = DT_F64 if out_dtype == torch.float64 else DT_F32
        self.step_mode = _MODES[step_mode]
        self._ccfg = cfg.to_c()
        st = self.lib.diral_env_validate(ctypes.byref(self._ccfg))
        if st != OK:
            raise DiralError(st, "diral_env_validate")
        self._h = ctypes.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        st = self.lib.diral_env_create(ctypes.byref(self._ccfg), self.B, dev_index, ctypes.byref(self._h))
        if st != OK:
            self._h = None
            raise DiralError(st, "diral_env_create")
        assert self.lib.diral_env_state_space(ctypes.byref(self._ccfg)) == self.S
        # I/O tensors are allocated once; step() never allocates
        if int(io_ring) < 1:
            raise ValueError("io_ring must be >= 1")
        self.io_ring = int(io_ring)
        if out_buffers is not None:
            # caller-owned output tensors (e.g. this handle's slice of a batch stepped as several sub-batches,
            # diral_amd/streamed.py): contiguous, right shape / dtype / device, io_ring == 1
            if self.io_ring != 1:
                raise ValueError("out_buffers needs io_ring == 1")
            want = dict(obs=((self.B, self.N, self.S), out_dtype), rew=((self.B, self.N), out_dtype),
                        done=((self.B,), torch.uint8), chobs=((self.B, self.N, self.CW), out_dtype))
            for k, (shape, dt) in want.items():
                tns = out_buffers.get(k)
                if tns is None and k == "chobs":
                    continue
                if tns is None or tuple(tns.shape) != shape or tns.dtype != dt or tns.device != self.device or not tns.is_contiguous():
                    raise ValueError("out_buffers[%r] must be a contiguous %s %s tensor on %s" % (k, shape, dt, self.device))
            self._ring = [dict(obs=out_buffers["obs"], rew=out_buffers["rew"], done=out_buffers["done"],
                               chobs=out_buffers.get("chobs"))]
        else:
            with torch.cuda.device(self.device):
                self._ring = [dict(obs=torch.zeros((self.B, self.N, self.S), dtype=out_dtype, device=self.device),
                                   rew=torch.zeros((self.B, self.N), dtype=out_dtype, device=self.device),
                                   done=torch.zeros((self.B,), dtype=torch.uint8, device=self.device), chobs=None)
                              for _ in range(self.io_ring)]
        self._ri = 0
        self._obs, self._rew, self._done = self._ring[0]["obs"], self._ring[0]["rew"], self._ring[0]["done"]
        self._chobs: Optional[torch.Tensor] = None
        self.t = 0
        self.env_offset = 0
        if env_offset:
            self.set_env_offset(env_offset)
        self.speculate_state = bool(speculate_state)
        self._spec: Optional[tuple] = None      # what the speculative state of the last my_step* is valid for
        self._vel_calls = 0                     # default-seed counter of update_velocity()
        self._clock: Optional[torch.Tensor] = None   # device slot clock installed by set_clock(); kept alive here

    # ---- lifetime -------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._clock = None
            h, self._h = self._h, None
            # `diral_env_destroy` is a series of hipFree calls: inside a stream capture (the cyclic collector may run this
            # finalizer at any allocation - also between two captured launches of ANOTHER handle) that is a fatal error of
            # the process.  A handle that dies during a capture is parked and destroyed with the next one that does not.
            if _capturing():
                _PARKED.append((self.lib, h))
                return
            self.lib.diral_env_destroy(h)
            while _PARKED:
                lib, ph = _PARKED.pop()
                lib.diral_env_destroy(ph)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers --------------------------------------------------------------
    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ok(self, st: int, where: str) -> None:
        if st != OK:
            detail = ""
            if st == ERR_HIP:
                detail = self.lib.diral_env_last_hip_error(self._h).decode()
            raise DiralError(st, where, detail)

    def _f64(self, a, shape) -> Optional[torch.Tensor]:
        if a is None:
            return None
        t = torch.as_tensor(a, dtype=torch.float64, device=self.device)
        return t.expand(shape).contiguous()

    def _actions(self, actions) -> torch.Tensor:
        a = torch.as_tensor(actions, device=self.device)
        if a.dtype != torch.int32:
            a = a.to(torch.int32)
        if a.dim() == 1:
            a = a.unsqueeze(0).expand(self.B, self.N)
        if tuple(a.shape) != (self.B, self.N):
            raise ValueError("actions must have shape [B=%d, N=%d], got %s" % (self.B, self.N, tuple(a.shape)))
        return a.contiguous()

    def hbm_bytes(self) -> int:
        return int(self.lib.diral_env_hbm_bytes(self._h))

    def set_env_offset(self, offset: int) -> None:
        """Global index of env 0 (sharded batches, see the class docstring)."""
        self._ok(self.lib.diral_env_set_option(self._h, OPT_ENV_OFFSET, int(offset)), "diral_env_set_option")
        self.env_offset = int(offset)

    def force_general_kernel(self, on: bool = True) -> None:
        """Tests / A-B timing: run every step on the general kernel (csrc/step_kernel.hpp)."""
        self._ok(self.lib.diral_env_set_option(self._h, OPT_KERNEL_PATH, PATH_GENERAL if on else PATH_AUTO),
                 "diral_env_set_option")

    def force_large_path(self, on: bool = True) -> None:
        """Tests: run every step / observe on the three launches of csrc/step_large.hpp, the form that serves
        num_users > 256, num_channels > 256 or num_bins > 64 (there it is the only path and this is a no-op)."""
        self._ok(self.lib.diral_env_set_option(self._h, OPT_KERNEL_PATH, PATH_LARGE if on else PATH_AUTO),
                 "diral_env_set_option")

    def set_clock(self, clock: Optional[torch.Tensor]) -> None:
        """Install (or, with None, remove) a device slot clock: an int64 tensor of one element on this env's device.
        While it is set the kernels read the slot number as ``t + clock[0]`` on the device (`done`, arrival stamps,
        trace replay) - what lets a captured hipGraph of K steps replay with a clock that moves on
        (`diral_env_set_clock`).  The env keeps a reference to the tensor: the C handle only stores its raw
        address, and a freed block would be handed out again by the caching allocator."""
        if clock is not None:
            if (not isinstance(clock, torch.Tensor) or clock.dtype != torch.int64 or clock.numel() != 1
                    or clock.device != self.device or not clock.is_contiguous()):
                raise ValueError("the slot clock must be a contiguous int64 tensor of one element on %s" % (self.device,))
        self._ok(self.lib.diral_env_set_clock(self._h, ctypes.c_void_p(clock.data_ptr()) if clock is not None else None),
                 "diral_env_set_clock")
        self._clock = clock

    def set_capture_rotation(self, on: bool) -> int:
        """Let captured step launches rotate the slow-env sets like eager ones (`diral_env_set_capture_rotation`; N <= 64).
        The caller's side of the contract: every graph captured while this is on holds a multiple of 3 step launches of
        this env, is replayed whole, and `align_phase(phase)` runs before a replay that follows anything but another
        replay of the same graph.  Returns the env's phase at the call (= the phase of the first launch captured next)."""
        phase = ctypes.c_int(0)
        self._ok(self.lib.diral_env_set_capture_rotation(self._h, 1 if on else 0, ctypes.byref(phase)),
                 "diral_env_set_capture_rotation")
        return int(phase.value)

    def align_phase(self, phase: int) -> None:
        """Bring the env's launch phase to `phase` on the current stream (`diral_env_align_phase`): a no-op when it is
        there already, otherwise the slow-env sets are emptied first."""
        self._ok(self.lib.diral_env_align_phase(self._h, int(phase), self._stream()), "diral_env_align_phase")

    def last_kernel(self) -> int:
        """config.KERNEL_* code of the kernel the last step / observe call launched."""
        return int(self.lib.diral_env_last_kernel(self._h))

    # ---- reference getters (test_env.py:486-496) ----------------------------
    def get_total_users(self) -> int:
        return self.N

    def get_num_ch(self) -> int:
        return self.A

    def get_state_space(self) -> int:
        return self.S

    def get_action_space(self) -> int:
        return self.A

    # ---- topology ---------------------------------------------------------------
    def reset_topology(self, x0=None, y0=None, v0=None, seed: int = 0) -> None:
        """New vehicles with zeroed tables: what constructing ``TestEnv`` does
        (network.py:92-119) or ``reset_mobility_env`` (test_env.py:479-484)
        when x0/y0/v0 are given."""
        shape = (self.B, self.N)
        x0, y0, v0 = self._f64(x0, shape), self._f64(y0, shape), self._f64(v0, shape)
        self._keep = (x0, y0, v0)
        self._ok(self.lib.diral_env_reset(self._h, _ptr(x0), _ptr(y0), _ptr(v0), int(seed) & (2**64 - 1),
                                          self._stream()), "diral_env_reset")
        self.t = 0
        self._spec = None
        self._vel_calls = 0                     # default-seed velocity draws after a reset == those of a fresh env

    def reset(self, x0=None, y0=None, v0=None, seed: int = 0, actions=None) -> torch.Tensor:
        """``reset() -> obs``.  The reference has no reset(); its driver
        bootstraps with ``action = env.sample(); obs, rews = env.my_step(action, 0);
        state = env.obtain_state(obs, action, rews)`` (main_test.py:89-94).
        This does exactly that on a fresh topology and returns the state."""
        self.reset_topology(x0, y0, v0, seed)
        a = self.sample(seed=seed + 0x5EED) if actions is None else self._actions(actions)
        obs, _, _ = self._step(STEP_MY_STEP, a, 0)
        self.t = 0
        return obs

    def reset_mobility_env(self) -> None:
        """test_env.py:479-484 -> network.py:81-90: the fixed 4-UE toy topology."""
        if self.N != 4:
            raise ConfigError("reset_mobility_env builds the fixed 4-UE topology (network.py:81-90)")
        self.reset_topology([3., 5., 3., 5.], [1., 1., 2., 2.], [0.5, 1.0, 1.25, 1.5])

    def reset_design_topology(self) -> None:
        """network.py:69-79: six vehicles 195 m apart, v = 1.0."""
        if self.N != 6:
            raise ConfigError("the design topology has 6 vehicles (network.py:69-79)")
        self.reset_topology([0., 195., 390., 585., 780., 975.], [1., 1., 2., 2., 2., 2.], [1.0] * 6)

    # ---- stepping ---------------------------------------------------------------
    def sample(self, seed: Optional[int] = None) -> torch.Tensor:
        """test_env.py:116-122: uniform random actions, [B, N] int32."""
        out = torch.empty((self.B, self.N), dtype=torch.int32, device=self.device)
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        self._ok(self.lib.diral_env_sample(self._h, _ptr(out), int(seed) & (2**64 - 1), self._stream()),
                 "diral_env_sample")
        return out

    def prefill(self, actions: torch.Tensor, slots: int, seed: int, rew_in=None, want_states: bool = True,
                episode: float = 0.0, eps: float = 1.0):
        """The driver's random prefill (main_test.py:99-114) as ONE launch of `slots` slots (`diral_env_prefill`):
        slot 0 runs `actions` ([B, N] int32, e.g. ``env.sample(seed)``), slot k the draw ``env.sample(seed + k)``
        would make; every slot is ``my_step_design(a_k, 0)`` followed by ``obtain_state(obs, a_k, rew_in)``.
        Returns ``(states [K, B, N, S] or None, actions_all [K, B, N], next_actions [B, N])``; `next_actions` is
        the draw of ``seed + slots`` (pass it, with that seed, to continue).  Bit-equal to the loop of
        sample + my_step_design + obtain_state calls.  Raises DiralError(ERR_UNSUPPORTED) with nothing launched for
        configurations the fused kernel does not take (diral_amd.driver.DriverLoop.prefill loops then)."""
        K = int(slots)
        a = self._actions(actions)
        states = torch.empty((K, self.B, self.N, self.S), dtype=self.out_dtype, device=self.device) if (want_states and self.S > 0) else None
        a_all = torch.empty((K, self.B, self.N), dtype=torch.int32, device=self.device)
        a_next = torch.empty((self.B, self.N), dtype=torch.int32, device=self.device)
        rin = None if rew_in is None else self._f64(rew_in, (self.B, self.N))
        self._spec = None
        st = self.lib.diral_env_prefill(self._h, _ptr(a), K, int(seed) & (2**64 - 1), _ptr(states) if states is not None else None,
                                        self._dt, _ptr(a_all), _ptr(a_next), _ptr(rin) if rin is not None else None,
                                        float(episode), float(eps), self._stream())
        self._ok(st, "diral_env_prefill")
        if rin is not None:
            torch.cuda.current_stream(self.device).synchronize()   # `rin` may be a temporary
        return states, a_all, a_next

    def _step(self, mode: int, actions: torch.Tensor, t: int, episode: float = 0.0, eps: float = 1.0,
              want_chobs: bool = False, want_obs: bool = True, stream: Optional[ctypes.c_void_p] = None):
        if self.io_ring > 1:
            self._ri = (self._ri + 1) % self.io_ring
        slot = self._ring[self._ri]
        if want_chobs and slot["chobs"] is None:
            slot["chobs"] = torch.zeros((self.B, self.N, self.CW), dtype=self.out_dtype, device=self.device)
        self._obs, self._rew, self._done, self._chobs = slot["obs"], slot["rew"], slot["done"], slot["chobs"]
        self._spec = None
        st = self.lib.diral_env_step(self._h, mode, _ptr(actions), int(t),
                                     _ptr(self._obs) if (want_obs and self.S > 0) else None,
                                     _ptr(self._rew), _ptr(self._done),
                                     _ptr(self._chobs) if want_chobs else None,
                                     self._dt, float(episode), float(eps), self._stream() if stream is None else stream)
        self._ok(st, "diral_env_step")
        return self._obs, self._rew, self._done

    def step_policy(self, actions: torch.Tensor, t: int, policy, actions_out: torch.Tensor, shaped_out=None, sum_r_out=None,
                    collision_out=None, global_reward_avg: bool = True, want_chobs: bool = False, clock=None,
                    seed_offset: Optional[int] = None, mode: Optional[int] = None, slots: int = 1, vel_seed: int = 0,
                    want_obs: bool = True, stuck_penalty: Optional[tuple] = None):
        """One slot of a policy-only rollout as ONE launch (`diral_env_step_policy`): env step (state, reward, done,
        optionally the channel observation) + the driver's reward shaping (main_test.py:171-206 without the
        information-age terms) + the SPS agents' decisions for the next slot (algorithms/v2x_sps.py:76-104), the
        channel observation handed over in LDS.  `policy`: a diral_amd.sps.SpsPolicy (its prev_action / counter are
        updated in place); `actions_out` [B, N] int32: the next slot's actions.  Same results, bit for bit, as
        `_step(want_chobs=True)` + `diral_driver_shape` + `SpsPolicy.step_from_chobs[_clocked]`; configurations the
        fused kernel does not take run exactly those three launches (then the channel observation is materialised
        whatever `want_chobs` says).  `clock`: a rollout.SlotClock / int64 device tensor added to the policy's seed
        (`step_from_chobs_clocked` semantics, `seed_offset` as its `offset`); without it the policy's own step counter
        advances as in `step_from_chobs`.

        `stuck_penalty` = (threshold, value, counter, prev_actions): the driver's penalty for an agent that repeats an
        unsuccessful action more than `threshold` times (main_test.py:194-203; `counter`, `prev_actions` [B, N] int32,
        updated in place - diral_driver_shape's flag bit 2); needs `shaped_out`.

        `slots` = K > 1: K slots in ONE launch, the env kept on the chip from slot to slot (include/diral_env.h,
        DiralSlotPolicy::slots): `actions` are slot t's, the policy decides the later ones; obs / reward / done / the channel
        observation are the LAST slot's (without `want_obs` no slot computes a state vector), `shaped_out` [K, B, N] and
        `sum_r_out` / `collision_out` [K, B] hold every slot's; configs with mobility_vary update the velocities at the
        episode ends inside the launch (`update_velocity(seed=vel_seed + slot // episode_interval)`).  Equal, bit for bit,
        to K one-slot calls; raises DiralError(UNSUPPORTED) where the fused kernel does not apply."""
        from .config import DiralSlotPolicy, ERR_UNSUPPORTED
        K = int(slots)
        if K < 1:
            raise ValueError("step_policy: slots must be >= 1")
        # raw pointers cross the C-ABI: what they point at is checked here (an int64 or strided tensor would be read as
        # garbage; `actions_out` aliasing `actions` lets the fused kernel's policy wave overwrite actions the
        # three-launch form still reads)
        for name, a in (("actions", actions), ("actions_out", actions_out)):
            if not isinstance(a, torch.Tensor) or a.dtype != torch.int32 or tuple(a.shape) != (self.B, self.N) \
                    or not a.is_contiguous() or a.device != self.device:
                raise ValueError("step_policy: %s must be a contiguous int32 tensor [%d, %d] on %s" %
                                 (name, self.B, self.N, self.device))
        if actions_out.data_ptr() == actions.data_ptr():
            raise ValueError("step_policy: actions_out must not alias actions")
        lead = (K,) if K > 1 else ()
        for name, a, shape in (("shaped_out", shaped_out, lead + (self.B, self.N)), ("sum_r_out", sum_r_out, lead + (self.B,)),
                               ("collision_out", collision_out, lead + (self.B,))):
            if a is not None and (a.dtype != self.out_dtype or tuple(a.shape) != shape or not a.is_contiguous()
                                  or a.device != self.device):
                raise ValueError("step_policy: %s must be a contiguous %s tensor %s on %s" %
                                 (name, self.out_dtype, shape, self.device))
        if self.io_ring > 1:
            self._ri = (self._ri + 1) % self.io_ring
        slot = self._ring[self._ri]
        # (whether the slot runs fused is the library's decision - kernel family, step mode, run-time extras; a call
        # without a channel-observation buffer that cannot run fused comes back DIRAL_ERR_UNSUPPORTED before anything
        # is launched, and is repeated with the buffer from then on)
        fusable = (self.N <= 64 and self.N >= 8 and self.A <= 64) and not getattr(self, "_policy_needs_chobs", False)
        if (want_chobs or not fusable) and slot["chobs"] is None:
            slot["chobs"] = torch.zeros((self.B, self.N, self.CW), dtype=self.out_dtype, device=self.device)
        self._obs, self._rew, self._done, self._chobs = slot["obs"], slot["rew"], slot["done"], slot["chobs"]
        self._spec = None
        q = DiralSlotPolicy()
        q.struct_bytes = ctypes.sizeof(DiralSlotPolicy)
        q.shape_flags = 1 if global_reward_avg else 0
        if stuck_penalty is not None:
            thr, val, cnt, prev = stuck_penalty
            for name, a in (("counter", cnt), ("prev_actions", prev)):
                if not isinstance(a, torch.Tensor) or a.dtype != torch.int32 or tuple(a.shape) != (self.B, self.N) \
                        or not a.is_contiguous() or a.device != self.device:
                    raise ValueError("step_policy: stuck_penalty %s must be a contiguous int32 tensor [%d, %d] on %s" %
                                     (name, self.B, self.N, self.device))
            if shaped_out is None:
                raise ValueError("step_policy: stuck_penalty needs shaped_out")
            q.shape_flags |= 4
            q.pen_threshold, q.pen_value = int(thr), float(val)
            q.pen_counter, q.pen_prev_actions = _ptr(cnt), _ptr(prev)
        q.shaped_out = _ptr(shaped_out); q.sum_r_out = _ptr(sum_r_out); q.collision_out = _ptr(collision_out)
        q.sps_prev_action = _ptr(policy.prev_action); q.sps_counter = _ptr(policy.counter)
        q.rssi_threshold, q.inc_db, q.keep_prob = policy.threshold, policy.inc_db, policy.keep_prob
        if clock is not None:
            ct = clock.t if hasattr(clock, "t") else clock
            q.seed = (int(policy.seed) * 1000003 + int(seed_offset or 0)) & (2**64 - 1)
            q.seed_clock = _ptr(ct)
        else:
            q.seed = (int(policy.seed) * 1000003 + policy._t + 1) & (2**64 - 1)
        q.actions_out = _ptr(actions_out)
        q.slots = K
        q.vel_seed = int(vel_seed) & (2**64 - 1)
        use_chobs = self._chobs if (want_chobs or not fusable) else None

        def call(chobs):
            return self.lib.diral_env_step_policy(self._h, self.step_mode if mode is None else mode, _ptr(actions), int(t),
                                                  _ptr(self._obs) if (want_obs and self.S > 0) else None, _ptr(self._rew),
                                                  _ptr(self._done), _ptr(chobs), self._dt, ctypes.byref(q), self._stream())
        st = call(use_chobs)
        if st == ERR_UNSUPPORTED and use_chobs is None and K == 1:  # not a fused configuration, nothing launched
            self._policy_needs_chobs = True
            if slot["chobs"] is None:
                slot["chobs"] = torch.zeros((self.B, self.N, self.CW), dtype=self.out_dtype, device=self.device)
            self._chobs = slot["chobs"]
            st = call(self._chobs)
        if st != OK and self.io_ring > 1:
            self._ri = (self._ri - 1) % self.io_ring            # nothing was launched: the ring slot is not consumed either
        self._ok(st, "diral_env_step_policy")
        if clock is None:
            policy._t += K        # only once the call is in: slot k drew with seed + k, what K one-slot calls are given; a
                                  # refused call (DIRAL_ERR_UNSUPPORTED: nothing launched) leaves the draw counter alone
        self._keep_policy = (q, actions, actions_out, shaped_out, sum_r_out, collision_out, clock, stuck_penalty)
        return self._obs, self._rew, self._done

    def step(self, actions, t: Optional[int] = None, episode: float = 0.0, epsilon: float = 1.0
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """``step(actions[B,N]) -> (obs[B,N,S], reward[B,N], done[B])``.

        One fused launch = the reference's per-slot pair
        ``obs, reward = env.my_step(action, t)`` +
        ``env.obtain_state(obs, action, reward, episode, eps)``
        (main_test.py:144-164).  ``done`` is the driver's episode boundary
        ``t % episode_interval == episode_interval-1`` (main_test.py:226); state is
        kept across it (continuing task).  The returned tensors are owned by the
        env and overwritten by the next call; enqueue-only, no host sync."""
        a = self._actions(actions)
        if t is None:
            t = self.t
        out = self._step(self.step_mode, a, t, episode, epsilon)
        self.t = int(t) + 1
        return out

    # reference-named batched variants: return (chobs[B,N,A], rews[B,N]) like
    # `obs, rews = env.my_step(actions, timestep)` and leave the state vector
    # to obtain_state(), so a main_test-style loop reads the same.
    def _ref_step(self, mode: int, actions, t: int):
        a = self._actions(actions)
        self._last_actions = a
        self._step(mode, a, t, want_chobs=True, want_obs=self.speculate_state)
        if self.speculate_state and self.S > 0:
            # the state vector built in the same launch is what obtain_state(chobs, actions, rew)
            # would build now; remember what it is valid for (tensor identity + version counters)
            given = actions if isinstance(actions, torch.Tensor) else None
            self._spec = (given, a, a._version, self._chobs._version, self._rew._version,
                          None if given is None else given._version)
        return self._chobs, self._rew

    def my_step(self, actions, timestep: int = 0):
        return self._ref_step(STEP_MY_STEP, actions, timestep)

    def my_step_ch(self, actions, time_step: int = 0):
        return self._ref_step(STEP_MY_STEP_CH, actions, time_step)

    def my_step_design(self, actions, timestep: int = 0):
        return self._ref_step(STEP_DESIGN, actions, timestep)

    def _spec_state_for(self, obs, acts, rewards, episode: float, eps: float) -> Optional[torch.Tensor]:
        """The speculative state of the last my_step* if `obtain_state(obs, acts, rewards)` asks
        for exactly that: the tensors my_step* returned, unmodified, and the actions it was given."""
        sp = self._spec
        if sp is None:
            return None
        given, a, a_ver, c_ver, r_ver, g_ver = sp
        st = self.cfg.State
        # `obs` / `rewards` only matter to the state through their own sections (test_env.py:540, 563)
        if st.add_channel_obs and (obs is not self._chobs or self._chobs._version != c_ver):
            return None
        if st.add_reward and (rewards is not self._rew or self._rew._version != r_ver):
            return None
        if a._version != a_ver:
            return None
        # (`a` may be a converted COPY of the caller's tensor: an in-place edit of that tensor between my_step
        # and obtain_state must miss, so its own version counter is part of the key)
        if not (acts is a or (given is not None and acts is given and given._version == g_ver)):
            t = torch.as_tensor(acts, device=self.device)
            if t.shape != a.shape and t.dim() == 1:
                t = t.unsqueeze(0).expand(self.B, self.N)
            if t.shape != a.shape or not bool(torch.equal(t.to(torch.int32), a)):
                return None
        if self.cfg.enable_fingerprint:
            # (episode, eps) columns: test_env.py:577-579 - the only part of the state that
            # obtain_state's extra arguments decide
            self._write_fingerprint(float(episode), float(eps))
        return self._obs

    def _fp_offset(self) -> int:
        # section order of obtain_state (test_env.py:527-583): ... velocity, fingerprint last
        return self.S - 2

    def _write_fingerprint(self, episode: float, eps: float) -> None:
        o = self._fp_offset()
        self._obs[:, :, o] = episode
        self._obs[:, :, o + 1] = eps

    def obtain_state(self, obs, acts, rewards, episode_number: float = 0, epsilon: float = 1) -> torch.Tensor:
        """test_env.py:527-583 on the current tables/positions; [B, N, S]."""
        hit = self._spec_state_for(obs, acts, rewards, episode_number, epsilon)
        if hit is not None:
            return hit
        self._spec = None
        a = self._actions(acts)
        chobs = None if obs is None else self._f64(obs, (self.B, self.N, self.CW))
        rew = None if rewards is None else self._f64(rewards, (self.B, self.N))
        st = self.lib.diral_env_observe(self._h, _ptr(a), _ptr(chobs), _ptr(rew), _ptr(self._obs), self._dt,
                                        float(episode_number), float(epsilon), self._stream())
        self._ok(st, "diral_env_observe")
        return self._obs

    def update_velocity(self, draws=None, seed: Optional[int] = None) -> None:
        """test_env.py:498-504 -> network.py:208-223 (no-op unless mobility_vary)."""
        d = None
        if draws is not None:
            d = torch.as_tensor(draws, dtype=torch.uint8, device=self.device).expand(self.B, self.N).contiguous()
        if seed is None:
            # a fresh draw per call (network.py:208-223 draws from the global RNG every episode):
            # the default seed counts update_velocity() calls, not env steps
            self._vel_calls += 1
            seed = self._vel_calls * 2654435761 + 12345
        self._spec = None
        self._ok(self.lib.diral_env_update_velocity(self._h, _ptr(d), int(seed) & (2**64 - 1), self._stream()),
                 "diral_env_update_velocity")

    def load_saved_positions(self, x_positions=None) -> None:
        """test_env.py:109-114 -> network.py:171-178: replay recorded x positions,
        `pos_x[u] = x_positions[t % T][u]` after every step.  `x_positions`: [T, N]
        (shared, the reference's `positions_*.npy` shape), [B, T, N] (per env), a
        path to such a .npy, or None to go back to the velocity model.  With no
        argument the config's `load_file_pos` is used when `load_positions` is set."""
        if x_positions is None and self.cfg.load_positions:
            x_positions = self.cfg.extra.get("load_file_pos")
        self._spec = None
        if x_positions is None:
            self._ok(self.lib.diral_env_set_trace(self._h, None, 0, 0, self._stream()), "diral_env_set_trace")
            return
        if isinstance(x_positions, str):
            import numpy as np
            x_positions = np.load(x_positions)
        tr = torch.as_tensor(x_positions, dtype=torch.float64, device=self.device).contiguous()
        if tr.dim() == 2 and tr.shape[1] == self.N:
            per_env, T = 0, tr.shape[0]
        elif tr.dim() == 3 and tuple(tr.shape[::2]) == (self.B, self.N):
            per_env, T = 1, tr.shape[1]
        else:
            raise ValueError("x_positions must be [T, N] or [B, T, N]")
        self._ok(self.lib.diral_env_set_trace(self._h, _ptr(tr), int(T), per_env, self._stream()),
                 "diral_env_set_trace")

    # ---- state access ---------------------------------------------------------
    def get_x_pos(self) -> torch.Tensor:
        """test_env.py:471-476, [B, N] float64."""
        return self.export_state(tables=False)["pos_x"]

    def export_state(self, tables: bool = True) -> Dict[str, torch.Tensor]:
        B, N = self.B, self.N
        f = dict(dtype=torch.float64, device=self.device)
        i = dict(dtype=torch.int32, device=self.device)
        out = dict(pos_x=torch.empty((B, N), **f), pos_y=torch.empty((B, N), **f), vel=torch.empty((B, N), **f))
        seq = age = x = y = la = None
        if tables:
            seq, age = torch.empty((B, N, N), **i), torch.empty((B, N, N), **i)
            x, y = torch.empty((B, N, N), **f), torch.empty((B, N, N), **f)
            out.update(seq=seq, age=age, x=x, y=y)
            if self.cfg.track_arrival:
                la = torch.empty((B, N, N), **i)
                out["la"] = la
        self._ok(self.lib.diral_env_export_state(self._h, _ptr(out["pos_x"]), _ptr(out["pos_y"]), _ptr(out["vel"]),
                                                 _ptr(seq), _ptr(age), _ptr(x), _ptr(y), _ptr(la), self._stream()),
                 "diral_env_export_state")
        return out

    def import_state(self, pos_x=None, pos_y=None, vel=None, seq=None, age=None, x=None, la=None) -> None:
        B, N = self.B, self.N
        def ti(a):
            return None if a is None else torch.as_tensor(a, dtype=torch.int32, device=self.device).expand(B, N, N).contiguous()
        self._spec = None
        px, py, v = self._f64(pos_x, (B, N)), self._f64(pos_y, (B, N)), self._f64(vel, (B, N))
        s, a, xx, l = ti(seq), ti(age), self._f64(x, (B, N, N)), ti(la)
        self._ok(self.lib.diral_env_import_state(self._h, _ptr(px), _ptr(py), _ptr(v), _ptr(s), _ptr(a), _ptr(xx),
                                                 _ptr(l), self._stream()), "diral_env_import_state")
        torch.cuda.current_stream(self.device).synchronize()   # inputs may be temporaries

    def export_entries(self) -> torch.Tensor:
        """The neighbour tables as RealNeS `MA_NeighborTableEntry` records (envs/ma_messages_pb2.py:195-230;
        realness_bridge.py:168-191): [B, N, N, 16] uint8 indexed [env, viewer, subject];
        `.cpu().numpy().view(ENTRY_DTYPE)[..., 0]` names the fields (pos_x, pos_y f32; seq_num, last_update i32)."""
        out = torch.empty((self.B, self.N, self.N, 16), dtype=torch.uint8, device=self.device)
        self._ok(self.lib.diral_env_export_entries(self._h, _ptr(out), self._stream()), "diral_env_export_entries")
        return out

    def import_entries(self, entries) -> None:
        """Load tables received in the RealNeS record layout (a [B, N, N] ENTRY_DTYPE array or the uint8
        tensor `export_entries` returns); pos_y is ignored, pos_x widens exactly to f64."""
        if isinstance(entries, np.ndarray):
            entries = torch.from_numpy(np.ascontiguousarray(entries).view(np.uint8).reshape(entries.shape[:3] + (16,)))
        rec = entries.to(device=self.device, dtype=torch.uint8).contiguous()
        if tuple(rec.shape) != (self.B, self.N, self.N, 16):
            raise ValueError("entries must be [B, N, N] records of 16 bytes, got %s" % (tuple(rec.shape),))
        self._spec = None
        self._ok(self.lib.diral_env_import_entries(self._h, _ptr(rec), self._stream()), "diral_env_import_entries")
        torch.cuda.current_stream(self.device).synchronize()   # `rec` may be a temporary

    def prev_obs(self) -> torch.Tensor:
        """State.piggybacking: TestEnv.prev_obs (test_env.py:76-79, 260-261), [B, N, A] float64."""
        out = torch.empty((self.B, self.N, self.A), dtype=torch.float64, device=self.device)
        self._ok(self.lib.diral_env_export_prev_obs(self._h, _ptr(out), self._stream()), "diral_env_export_prev_obs")
        return out

    def set_prev_obs(self, prev_obs) -> None:
        p = self._f64(prev_obs, (self.B, self.N, self.A))
        self._spec = None
        self._ok(self.lib.diral_env_import_prev_obs(self._h, _ptr(p), self._stream()), "diral_env_import_prev_obs")
        torch.cuda.current_stream(self.device).synchronize()   # `p` may be a temporary

    def info_age(self, t: int) -> torch.Tensor:
        """network.py:560-574 (`env.network.get_information_age(t)`), [B, 100] int32."""
        out = torch.empty((self.B, 100), dtype=torch.int32, device=self.device)
        self._ok(self.lib.diral_env_info_age(self._h, int(t), _ptr(out), self._stream()), "diral_env_info_age")
        return out

    def metrics(self, clear: bool = False) -> torch.Tensor:
        """[B, 6] float64: slots, sum reward, sole tx, collided tx, PRR sum, PRR count."""
        out = torch.empty((self.B, M_COLUMNS), dtype=torch.float64, device=self.device)
        self._ok(self.lib.diral_env_metrics(self._h, _ptr(out), int(clear), self._stream()), "diral_env_metrics")
        return out

    def check(self) -> None:
        """Raise if a kernel flagged an out-of-range action, a sequence overflow or (State.piggybacking) a receiver
        without a transmitter in range since the last check (synchronises the stream)."""
        self._ok(self.lib.diral_env_check(self._h, self._stream()), "diral_env_check")
