"""SPS baseline policy on the device.

The reference ships 3GPP semi-persistent scheduling as ``SemiPersistentScheduling``
(algorithms/v2x_sps.py): one Python object per agent, fed an averaged-RSSI
"selection window" of length A.  :class:`SpsPolicy` keeps its decision logic
(reselection counter 5..15/16, keep-probability 0.8, pick among the lowest-RSSI
20 % after raising the threshold in 3 dB steps) for B x N agents in one launch of
``diral_sps_step`` (include/diral_env.h).  The reference never wires SPS to the
toy env, so :func:`rssi_from_channel_obs` - the map from the env's channel
observation to an RSSI-like vector - is a build extension, not reference
behaviour.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .vec_env import DiralError


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class SpsPolicy:
    def __init__(self, batch: int, num_users: int, num_channels: int, rssi_threshold: float = -110.0,
                 device="cuda:0", seed: int = 0, selection_window: Optional[int] = None):
        self.lib = _lib.load()
        self.B, self.N, self.A = batch, num_users, num_channels
        self.device = torch.device(device)
        self.threshold = float(rssi_threshold)          # v2x_sps.py:12
        self.inc_db = 3.0                               # v2x_sps.py:18
        self.keep_prob = 0.8                            # v2x_sps.py:22
        self.prev_action = torch.zeros((batch, num_users), dtype=torch.int32, device=self.device)
        self.counter = torch.zeros((batch, num_users), dtype=torch.int32, device=self.device)
        self._t = 0
        # v2x_sps.py:14: randint(0, selection_window) is inclusive; A-1 keeps actions in range
        window = num_channels - 1 if selection_window is None else selection_window
        st = self.lib.diral_sps_init(batch * num_users, int(window), _ptr(self.prev_action), _ptr(self.counter),
                                     int(seed) & (2**64 - 1), self._stream())
        if st != 0:
            raise DiralError(st, "diral_sps_init")
        self.seed = seed

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def step(self, selection_window: torch.Tensor, draw_counter=None, draw_keep=None, draw_choice=None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """selection_window [B, N, A] -> actions [B, N] int32 (v2x_sps.py:76-104)."""
        w = selection_window.to(device=self.device, dtype=torch.float64).contiguous()
        if tuple(w.shape) != (self.B, self.N, self.A):
            raise ValueError("selection_window must be [B, N, A]")
        if out is None:
            out = torch.empty((self.B, self.N), dtype=torch.int32, device=self.device)

        def opt(t, dt):
            return None if t is None else torch.as_tensor(t, dtype=dt, device=self.device).reshape(self.B, self.N).contiguous()
        dc, dk, dch = opt(draw_counter, torch.int32), opt(draw_keep, torch.float64), opt(draw_choice, torch.int32)
        self._t += 1
        st = self.lib.diral_sps_step(self.B * self.N, self.A, _ptr(w), _ptr(self.prev_action), _ptr(self.counter),
                                     self.threshold, self.inc_db, self.keep_prob, _ptr(dc), _ptr(dk), _ptr(dch),
                                     (int(self.seed) * 1000003 + self._t) & (2**64 - 1), _ptr(out), self._stream())
        if st != 0:
            raise DiralError(st, "diral_sps_step")
        self._keep = (w, dc, dk, dch)
        return out


    def _draws(self, draw_counter, draw_keep, draw_choice):
        def opt(t, dt):
            return None if t is None else torch.as_tensor(t, dtype=dt, device=self.device).reshape(self.B, self.N).contiguous()
        return opt(draw_counter, torch.int32), opt(draw_keep, torch.float64), opt(draw_choice, torch.int32)

    def window_from_chobs(self, chobs: torch.Tensor, actions: torch.Tensor) -> torch.Tensor:
        """`diral_sps_window_from_chobs`: the RSSI-like window [B, N, A] float64 from the env's channel
        observation (float32 or float64) and the actions that produced it (build extension, see
        include/diral_env.h)."""
        c = chobs.contiguous()
        a = actions.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty((self.B, self.N, self.A), dtype=torch.float64, device=self.device)
        st = self.lib.diral_sps_window_from_chobs(self.B * self.N, self.A, _ptr(c), 1 if c.dtype == torch.float64 else 0,
                                                  _ptr(a), _ptr(out), self._stream())
        if st != 0:
            raise DiralError(st, "diral_sps_window_from_chobs")
        return out

    def step_from_chobs(self, chobs: torch.Tensor, actions: torch.Tensor, draw_counter=None, draw_keep=None,
                        draw_choice=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One SPS step straight from the env's channel observation [B, N, A] (float32 / float64) and
        the actions of the slot that produced it: `window_from_chobs` + `step` in ONE launch, with the
        window built only for the agents that re-select (`diral_sps_step_chobs`).  A <= 256."""
        if chobs.dtype not in (torch.float32, torch.float64) or tuple(chobs.shape) != (self.B, self.N, self.A):
            raise ValueError("chobs must be float32/float64 [B, N, A]")
        c = chobs.contiguous()
        a = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if out is None:
            out = torch.empty((self.B, self.N), dtype=torch.int32, device=self.device)
        dc, dk, dch = self._draws(draw_counter, draw_keep, draw_choice)
        self._t += 1
        st = self.lib.diral_sps_step_chobs(self.B * self.N, self.A, _ptr(c), 1 if c.dtype == torch.float64 else 0, _ptr(a),
                                           _ptr(self.prev_action), _ptr(self.counter), self.threshold, self.inc_db,
                                           self.keep_prob, _ptr(dc), _ptr(dk), _ptr(dch),
                                           (int(self.seed) * 1000003 + self._t) & (2**64 - 1), _ptr(out), self._stream())
        if st != 0:
            raise DiralError(st, "diral_sps_step_chobs")
        self._keep = (c, a, dc, dk, dch)
        return out

    def step_from_chobs_clocked(self, chobs: torch.Tensor, actions: torch.Tensor, clock, offset: int = 0,
                                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`step_from_chobs` with device draws seeded by (seed, offset) + the value of a device slot counter
        (`diral_sps_step_chobs_clocked`): the by-value arguments of a captured launch stay fixed while the draws move on
        with the counter (diral_amd/rollout.py).  `clock`: a rollout.SlotClock or an int64 device tensor."""
        if chobs.dtype not in (torch.float32, torch.float64) or tuple(chobs.shape) != (self.B, self.N, self.A):
            raise ValueError("chobs must be float32/float64 [B, N, A]")
        c = chobs.contiguous()
        a = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if out is None:
            out = torch.empty((self.B, self.N), dtype=torch.int32, device=self.device)
        ct = clock.t if hasattr(clock, "t") else clock
        st = self.lib.diral_sps_step_chobs_clocked(self.B * self.N, self.A, _ptr(c), 1 if c.dtype == torch.float64 else 0, _ptr(a),
                                                   _ptr(self.prev_action), _ptr(self.counter), self.threshold, self.inc_db,
                                                   self.keep_prob, (int(self.seed) * 1000003 + int(offset)) & (2**64 - 1),
                                                   _ptr(ct), _ptr(out), self._stream())
        if st != 0:
            raise DiralError(st, "diral_sps_step_chobs_clocked")
        self._keep = (c, a, ct)
        return out


def rssi_from_channel_obs(chobs: torch.Tensor, actions: torch.Tensor) -> torch.Tensor:
    """Build extension: an RSSI-like selection window from the toy env's type-2
    channel observation (`obs[user][i]` = distance to the nearest in-range
    transmitter, 100000 if all are out of range, 0 if nobody transmitted or the
    agent transmitted there itself; test_env.py:206, 240, network.py:385).
    Log-distance path loss, lower = quieter; the agent's own resource reads as busy.
    (Plain-torch statement of `diral_sps_window_from_chobs`; `SpsPolicy.step_from_chobs` is the
    fused device path.)"""
    d = chobs.to(torch.float64)
    rssi = torch.full_like(d, -200.0)                                   # idle resource
    heard = (d > 0) & (d < 100000.0)
    rssi = torch.where(heard, -40.0 - 30.0 * torch.log10(torch.clamp(d, min=1.0)), rssi)
    rssi = torch.where(d >= 100000.0, torch.full_like(d, -160.0), rssi)   # busy, out of range
    own = torch.nn.functional.one_hot(actions.long(), d.shape[-1]).bool()
    return torch.where(own, torch.full_like(d, -60.0), rssi)
