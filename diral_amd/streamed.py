"""StreamedVecEnv - one batch of B envs stepped as G independent sub-batches on G HIP streams.

Envs never interact (one `TestEnv` per process in the reference, main_test.py:46), so slot t + 1 of a sub-batch
depends on slot t of the SAME sub-batch only.  One launch over all B envs cannot use that: it has to drain
completely before the next one starts, and at the headline batch (4096 workgroups on 256 CUs x 7 resident = 2.29
rounds) the last, mostly empty round costs a fifth of the slot.  With the batch cut into G sub-batches, each a
`VecV2VEnv` handle (its slice of ONE global batch: `env_offset`) on a stream of its own, the tail of one launch
overlaps with the head of the next: 70.7 -> 55.6 us per slot at G = 2, 52.6 us at G = 4 (c2, MI355X,
profiles/two_streams.py).

Outputs are ONE set of tensors [B, N, ...]; every sub-batch writes its slice.  `step()` enqueues G launches and,
by default (`sync=True`), makes the caller's stream wait for all of them - the single-handle semantics.  A slot loop
that can keep the sub-batches apart (a policy evaluated per sub-batch, pre-generated actions, the reference's
prefill phase with random actions, main_test.py:99-114) steps with `sync=False` and calls `wait()` when it needs
the outputs on its own stream; that is where the overlap comes from.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import torch

from .config import EnvConfig
from .shard import env_shard
from .vec_env import VecV2VEnv


class StreamedVecEnv:
    def __init__(self, cfg, batch: int, groups: int = 4, device="cuda:0", out_dtype: torch.dtype = torch.float32,
                 step_mode="my_step", env_offset: int = 0, want_chobs: bool = True):
        if groups < 1 or groups > batch:
            raise ValueError("groups must be in [1, batch]")
        if not isinstance(cfg, EnvConfig):
            cfg = EnvConfig.from_dict(cfg)
        self.cfg, self.B, self.G = cfg, int(batch), int(groups)
        self.device = torch.device(device)
        self.N, self.A, self.S = cfg.num_users, cfg.num_channels, cfg.state_space
        self.want_chobs = bool(want_chobs)
        with torch.cuda.device(self.device):
            self.obs = torch.zeros((self.B, self.N, self.S), dtype=out_dtype, device=self.device)
            self.rew = torch.zeros((self.B, self.N), dtype=out_dtype, device=self.device)
            self.done = torch.zeros((self.B,), dtype=torch.uint8, device=self.device)
            self.chobs = torch.zeros((self.B, self.N, cfg.chobs_width), dtype=out_dtype, device=self.device) if want_chobs else None
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.G)]
        self.slices: List[Tuple[int, int]] = []
        self.envs: List[VecV2VEnv] = []
        for g in range(self.G):
            start, count = env_shard(self.B, g, self.G)
            self.slices.append((start, count))
            sl = slice(start, start + count)
            bufs = dict(obs=self.obs[sl], rew=self.rew[sl], done=self.done[sl], chobs=self.chobs[sl] if want_chobs else None)
            self.envs.append(VecV2VEnv(cfg, batch=count, device=self.device, out_dtype=out_dtype, step_mode=step_mode,
                                       env_offset=env_offset + start, out_buffers=bufs))
        self.step_mode = self.envs[0].step_mode
        self._ready = torch.cuda.Event()
        self._raw = [ctypes.c_void_p(s.cuda_stream) for s in self.streams]     # the launches take the stream handle directly
        self._views = (None, None)      # (actions tensor the views were cut from, its per-sub-batch views)
        self.t = 0

    # ---- topology ---------------------------------------------------------------
    def reset_topology(self, x0=None, y0=None, v0=None, seed: int = 0) -> None:
        """As VecV2VEnv.reset_topology; device draws are indexed by the GLOBAL env index, so G sub-batches draw what
        one handle holding all B envs draws."""
        def part(a, sl):
            if a is None:
                return None
            t = torch.as_tensor(a)
            return t[sl] if t.dim() >= 2 and t.shape[0] == self.B else t
        for (start, count), e, s in zip(self.slices, self.envs, self.streams):
            sl = slice(start, start + count)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                e.reset_topology(part(x0, sl), part(y0, sl), part(v0, sl), seed)
        self.wait()
        self.t = 0

    def sample(self, seed: int = 0) -> torch.Tensor:
        out = torch.empty((self.B, self.N), dtype=torch.int32, device=self.device)
        for (start, count), e in zip(self.slices, self.envs):
            out[start:start + count] = e.sample(seed=seed)
        return out

    # ---- stepping ---------------------------------------------------------------
    def step(self, actions: torch.Tensor, t: Optional[int] = None, sync: bool = True, actions_ready: bool = False):
        """One slot of all B envs: G launches, one per sub-batch, each on its own stream behind an event of the
        caller's stream (the actions are ready).  `actions` [B, N] int32 on the device.  Returns (obs, reward,
        done) - and fills `self.chobs` - as ONE set of [B, ...] tensors; with sync=False they belong to the
        sub-batch streams until wait().  `actions_ready=True`: the caller guarantees that `actions` is complete
        (e.g. pre-generated and synchronised long ago) - the event hand-shake with the caller's stream is skipped."""
        if t is None:
            t = self.t
        converted = False
        if actions.dtype != torch.int32 or not actions.is_contiguous() or tuple(actions.shape) != (self.B, self.N):
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous().view(self.B, self.N)
            converted = True
            actions_ready = False                  # produced on the caller's stream just now: the hand-shake is needed
        if self._views[0] is not actions:
            self._views = (actions, [actions[start:start + count] for start, count in self.slices])
        views = self._views[1]
        if not actions_ready:
            self._ready.record(torch.cuda.current_stream(self.device))
        for g in range(self.G):
            if not actions_ready:
                self.streams[g].wait_event(self._ready)
            self.envs[g]._step(self.step_mode, views[g], t, want_chobs=self.want_chobs, stream=self._raw[g])
            if converted:
                # a tensor this call allocated on the caller's stream and the sub-batch streams read: without the
                # note the caching allocator may hand its block out again (on the caller's stream) while a launch
                # of a sub-batch stream is still reading it
                actions.record_stream(self.streams[g])
        self.t = int(t) + 1
        if sync:
            self.wait()
        return self.obs, self.rew, self.done

    def wait(self) -> None:
        """The caller's stream waits for every sub-batch stream (no host sync)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def update_velocity(self, draws=None, seed: Optional[int] = None) -> None:
        """As VecV2VEnv.update_velocity, per sub-batch on its stream: behind the caller's stream (which produced
        `draws`), and joined afterwards (the draws tensor may be released by the caller right after the call)."""
        cur = torch.cuda.current_stream(self.device)
        d = None if draws is None else torch.as_tensor(draws)
        for (start, count), e, s in zip(self.slices, self.envs, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.update_velocity(None if d is None else d[start:start + count], seed)
        self.wait()

    def metrics(self, clear: bool = False) -> torch.Tensor:
        self.wait()
        return torch.cat([e.metrics(clear=clear) for e in self.envs], dim=0)

    def export_state(self, tables: bool = True):
        self.wait()
        parts = [e.export_state(tables) for e in self.envs]
        return {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}

    def check(self) -> None:
        self.wait()
        for e in self.envs:
            e.check()

    def close(self) -> None:
        for e in self.envs:
            e.close()
