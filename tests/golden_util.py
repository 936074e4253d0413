"""Shared helpers: load tests/golden/*.npz and replay them through a backend.

A fixture holds inputs (config JSON, x0/y0/v0, per-step mode/actions/t, velocity
draws) and the reference's outputs for the same (rews, chobs, state, positions,
table planes).  tests/golden/gen_golden.py recorded them from the reference.
"""
import glob
import json
import os

import numpy as np

from diral_amd.config import EnvConfig, STEP_DESIGN, STEP_MY_STEP, STEP_MY_STEP_CH

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = {"step": STEP_MY_STEP, "ch": STEP_MY_STEP_CH, "design": STEP_DESIGN}


def golden_names(prefix="g"):
    """step-level fixtures are g*.npz; driver-loop fixtures (main_test.py call sequence) are d*.npz"""
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


class Golden:
    def __init__(self, name):
        self.name = name
        self.d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.cfg_dict = json.loads(str(self.d["cfg"]))
        self.cfg = EnvConfig.from_dict(self.cfg_dict, track_arrival=True)
        self.N = self.cfg.num_users
        self.A = self.cfg.num_channels
        self.T = len(self.d["modes"])
        self.vel_updates = {int(s): self.d["vel_update_draws"][i]
                            for i, s in enumerate(self.d["vel_update_steps"])}
        # trace replay fixtures: the recorded [T, N] trace and the step after which
        # the reference called load_saved_positions()
        self.trace = self.d["trace"] if "trace" in self.d and self.d["trace"].shape[0] else None
        self.trace_after = int(self.d["trace_after"]) if "trace_after" in self.d else -1

    def __getitem__(self, k):
        return self.d[k]

    def steps(self):
        for i in range(self.T):
            yield (i, MODES[str(self.d["modes"][i])], self.d["actions"][i],
                   int(self.d["tsteps"][i]), tuple(self.d["episode_eps"][i]))

    def table_checkpoints(self):
        if "tab_step" not in self.d or len(self.d["tab_step"]) == 0:
            return {}
        return {int(s): j for j, s in enumerate(self.d["tab_step"])}


def ulp_diff(a, b):
    """max distance in units-in-the-last-place between two float64 arrays."""
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    ai = a.view(np.int64).copy()
    bi = b.view(np.int64).copy()
    ai[ai < 0] = np.int64(-2**63) - ai[ai < 0]
    bi[bi < 0] = np.int64(-2**63) - bi[bi < 0]
    if a.size == 0:
        return 0
    return int(np.max(np.abs(ai - bi)))
