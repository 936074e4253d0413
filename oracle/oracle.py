"""ctypes binding of oracle/libdiral_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py - never by diral_amd/.  See diral_oracle.c.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from diral_amd.config import DiralCfg, EnvConfig, M_COLUMNS

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdiral_oracle.so")
_NATIVE_PATH = os.path.join(_HERE, "libdiral_oracle_native.so")
NATIVE_FLAGS = "-O3 -march=native -fPIC -ffp-contract=off -fno-fast-math -std=c11 -fopenmp"
CHECKER_FLAGS = "-O2 -fPIC -ffp-contract=off -fno-fast-math -std=c11 -fopenmp"
_lib = None
_native = None

SQ_POW = 0    # reference-faithful: v**2 == libm pow(v, 2.0)
SQ_IEEE = 1   # v*v, what the HIP kernels compute


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "diral_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "diral_env.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdiral_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def build_native() -> str:
    """The -O3 -march=native build of the same source (oracle/Makefile): what bench.py's `cpu_baseline` leg times
    (BASELINE.md section 4).  -march=native: always recompiled on the box it runs on."""
    subprocess.check_call(["make", "-C", _HERE, "-B", "libdiral_oracle_native.so"], stdout=subprocess.DEVNULL)
    return _NATIVE_PATH


def _load(native: bool = False):
    global _lib, _native
    if native:
        if _native is None:
            _native = _bind(ctypes.CDLL(build_native()))
        return _native
    if _lib is not None:
        return _lib
    build()
    _lib = _bind(ctypes.CDLL(_LIB_PATH))
    return _lib


def _bind(lib):
    P = ctypes.c_void_p
    lib.oracle_create.restype = P
    lib.oracle_create.argtypes = [ctypes.POINTER(DiralCfg), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.oracle_destroy.argtypes = [P]
    lib.oracle_reset.argtypes = [P, P, P, P]
    lib.oracle_step.argtypes = [P, ctypes.c_int, P, ctypes.c_int64, P, P]
    lib.oracle_step.restype = ctypes.c_int
    lib.oracle_prev_obs.argtypes = [P, P]
    lib.oracle_obtain_state.argtypes = [P, P, P, P, ctypes.c_double, ctypes.c_double, P]
    lib.oracle_update_velocity.argtypes = [P, P]
    lib.oracle_info_age.argtypes = [P, ctypes.c_int64, P]
    lib.oracle_export.argtypes = [P] + [P] * 9
    lib.oracle_import.argtypes = [P] + [P] * 8
    lib.oracle_metrics.argtypes = [P, P]
    lib.oracle_set_trace.argtypes = [P, P, ctypes.c_int]
    lib.oracle_edges.argtypes = [P, P]
    lib.oracle_state_space.argtypes = [ctypes.POINTER(DiralCfg)]
    lib.oracle_state_space.restype = ctypes.c_int
    lib.oracle_has_openmp.restype = ctypes.c_int
    return lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    """B independent reference-faithful envs on the CPU."""

    def __init__(self, cfg: EnvConfig, batch: int = 1, sq_mode: int = SQ_POW, threads: int = 1, native: bool = False):
        self.lib = _load(native)                # native: the -O3 -march=native build (bench.py's timed CPU leg only)
        self.cfg = cfg
        self.ccfg = cfg.to_c()
        self.B, self.N, self.A = batch, cfg.num_users, cfg.num_channels
        self.K = cfg.State.num_bins
        self.S = self.lib.oracle_state_space(ctypes.byref(self.ccfg))
        assert self.S == cfg.state_space
        self.threads = threads
        self.h = self.lib.oracle_create(ctypes.byref(self.ccfg), batch, sq_mode, threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_destroy(self.h)
            self.h = None

    def reset(self, x0, y0, v0) -> None:
        x0, y0, v0 = (np.ascontiguousarray(np.broadcast_to(a, (self.B, self.N)), dtype=np.float64)
                      for a in (x0, y0, v0))
        self.lib.oracle_reset(self.h, _p(x0), _p(y0), _p(v0))

    def step(self, mode: int, actions, t: int) -> Tuple[np.ndarray, np.ndarray]:
        a = np.ascontiguousarray(np.broadcast_to(actions, (self.B, self.N)), dtype=np.int32)
        rews = np.empty((self.B, self.N), np.float64)
        # State.piggybacking: my_step returns piggy_obs, A * A values per agent (test_env.py:263-264)
        cw = self.cfg.chobs_width if mode == 0 else self.A
        chobs = np.empty((self.B, self.N, cw), np.float64)
        if self.lib.oracle_step(self.h, mode, _p(a), int(t), _p(rews), _p(chobs)):
            raise KeyError(None)            # the reference's `self.prev_obs[tx_id]` with tx_id None (test_env.py:243)
        return rews, chobs

    def prev_obs(self) -> np.ndarray:
        """TestEnv.prev_obs (test_env.py:76-79, 260-261), [B, N, A]."""
        out = np.empty((self.B, self.N, self.A), np.float64)
        self.lib.oracle_prev_obs(self.h, _p(out))
        return out

    def obtain_state(self, actions, chobs, rews, episode: float = 0, eps: float = 1) -> np.ndarray:
        a = np.ascontiguousarray(np.broadcast_to(actions, (self.B, self.N)), dtype=np.int32)
        chobs = np.ascontiguousarray(chobs, dtype=np.float64).reshape(self.B, self.N, self.cfg.chobs_width)
        rews = np.ascontiguousarray(rews, dtype=np.float64).reshape(self.B, self.N)
        state = np.empty((self.B, self.N, self.S), np.float64)
        self.lib.oracle_obtain_state(self.h, _p(a), _p(chobs), _p(rews), float(episode),
                                     float(eps), _p(state))
        return state

    def update_velocity(self, draws) -> None:
        d = np.ascontiguousarray(np.broadcast_to(draws, (self.B, self.N)), dtype=np.uint8)
        self.lib.oracle_update_velocity(self.h, _p(d))

    def info_age(self, t: int) -> np.ndarray:
        out = np.empty((self.B, 100), np.int32)
        self.lib.oracle_info_age(self.h, int(t), _p(out))
        return out

    def export(self) -> dict:
        B, N = self.B, self.N
        d = dict(pos_x=np.empty((B, N)), pos_y=np.empty((B, N)), vel=np.empty((B, N)),
                 seq=np.empty((B, N, N), np.int32), age=np.empty((B, N, N), np.int32),
                 x=np.empty((B, N, N)), y=np.empty((B, N, N)),
                 la=np.empty((B, N, N), np.int64), pf=np.empty((B, N), np.int32))
        self.lib.oracle_export(self.h, *[_p(d[k]) for k in
                                         ("pos_x", "pos_y", "vel", "seq", "age", "x", "y", "la", "pf")])
        return d

    def import_state(self, pos_x=None, pos_y=None, vel=None, seq=None, age=None,
                     x=None, y=None, la=None) -> None:
        def c(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt)
        arrs = [c(pos_x, np.float64), c(pos_y, np.float64), c(vel, np.float64),
                c(seq, np.int32), c(age, np.int32), c(x, np.float64), c(y, np.float64),
                c(la, np.int64)]
        self.lib.oracle_import(self.h, *[_p(a) for a in arrs])

    def set_trace(self, trace) -> None:
        """Network.load_x_positions (network.py:171-178): [T, N] x-positions."""
        if trace is None:
            self.lib.oracle_set_trace(self.h, None, 0)
            return
        tr = np.ascontiguousarray(trace, dtype=np.float64).reshape(-1, self.N)
        self.lib.oracle_set_trace(self.h, _p(tr), tr.shape[0])

    def metrics(self) -> np.ndarray:
        out = np.empty((self.B, M_COLUMNS), np.float64)
        self.lib.oracle_metrics(self.h, _p(out))
        return out

    def edges(self) -> np.ndarray:
        out = np.empty(self.K + 1, np.float64)
        self.lib.oracle_edges(self.h, _p(out))
        return out


def has_openmp() -> bool:
    return bool(_load().oracle_has_openmp())
