"""TEST-ONLY stand-in for VecV2VEnv backed by the CPU oracle, so the host-side
logic that sits above the C-ABI (the reference-shaped ``TestEnv`` shim) can be
exercised on a box without a GPU.  Never imported by diral_amd/."""
import numpy as np

from diral_amd.config import STEP_DESIGN, STEP_MY_STEP, STEP_MY_STEP_CH
from oracle.oracle import Oracle, SQ_POW


class OracleBackend:
    def __init__(self, cfg, batch=1):
        self.cfg, self.B, self.N, self.A = cfg, batch, cfg.num_users, cfg.num_channels
        self.o = Oracle(cfg, batch=batch, sq_mode=SQ_POW)

    def get_total_users(self):
        return self.N

    def get_action_space(self):
        return self.A

    def get_state_space(self):
        return self.cfg.state_space

    def reset_topology(self, x0=None, y0=None, v0=None, seed=0):
        rng = np.random.default_rng(seed)
        L = int(self.cfg.highway_length)
        x0 = rng.integers(0, L, size=(self.B, self.N)).astype(float) if x0 is None else x0
        y0 = np.zeros((self.B, self.N)) if y0 is None else y0
        if v0 is None:
            v0 = np.full((self.B, self.N), 1.7) if self.cfg.mobility_vary else rng.uniform(1.1, 2.7, (self.B, self.N))
        self.o.reset(x0, y0, v0)

    def reset_mobility_env(self):
        self.reset_topology([3., 5., 3., 5.], [1., 1., 2., 2.], [0.5, 1.0, 1.25, 1.5])

    def reset_design_topology(self):
        self.reset_topology([0., 195., 390., 585., 780., 975.], [1., 1., 2., 2., 2., 2.], [1.0] * 6)

    def sample(self, seed=None):
        return np.random.default_rng(seed).integers(0, self.A, size=(self.B, self.N)).astype(np.int32)

    def _step(self, mode, a, t):
        rews, chobs = self.o.step(mode, a, t)
        return chobs, rews

    def my_step(self, a, t=0):
        return self._step(STEP_MY_STEP, a, t)

    def my_step_ch(self, a, t=0):
        return self._step(STEP_MY_STEP_CH, a, t)

    def my_step_design(self, a, t=0):
        return self._step(STEP_DESIGN, a, t)

    def obtain_state(self, obs, acts, rewards, episode_number=0, epsilon=1):
        return self.o.obtain_state(acts, obs, rewards, episode_number, epsilon)

    def update_velocity(self, draws=None, seed=None):
        if draws is None:
            draws = np.random.default_rng(seed).integers(1, 4, size=(self.B, self.N))
        self.o.update_velocity(draws)

    def load_saved_positions(self, x_positions=None):
        if isinstance(x_positions, str):
            x_positions = np.load(x_positions)
        self.o.set_trace(x_positions)

    def get_x_pos(self):
        return self.o.export()["pos_x"]

    def export_state(self, tables=True):
        return self.o.export()

    def info_age(self, t):
        return self.o.info_age(t)

    def check(self):
        pass            # (the oracle raises inside step itself, like the reference)
