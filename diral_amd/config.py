"""Env configuration: the reference's `EnvironmentTest` YAML block as one object.

The reference passes the YAML dict straight into ``TestEnv(**cfg)`` and reads
it with ``kwargs.setdefault`` (envs/test_env.py:12-48; nested ``State`` block
:26-41).  :class:`EnvConfig` accepts the same dict (or a reference YAML file)
verbatim, keeps the same defaults, and lowers to the C-ABI ``DiralCfg`` struct
(include/diral_env.h).
"""
from __future__ import annotations

import ctypes
import dataclasses
from dataclasses import dataclass, field
from typing import Any, Dict, Mapping, Optional

# ---- C-ABI mirror (include/diral_env.h) ------------------------------------

ABI_VERSION = 8

F_MOBILITY = 1 << 0
F_MOBILITY_VARY = 1 << 1
F_TOY_WEIGHTS = 1 << 2
F_ADD_ACTION = 1 << 3
F_ACTION_REAL = 1 << 4
F_ADD_CHANNEL_OBS = 1 << 5
F_ADD_REWARD = 1 << 6
F_ADD_INDEX = 1 << 7
F_ADD_VELOCITY = 1 << 8
F_ADD_POSITION = 1 << 9
F_ADD_POSDIST = 1 << 10
F_ADD_POSDIST_PIGGY = 1 << 11
F_FINGERPRINT = 1 << 12
F_PROPORTIONAL_FAIR = 1 << 13
F_DESIGN_TOPOLOGY = 1 << 14
F_PIGGYBACKING = 1 << 15
F_TRACK_ARRIVAL = 1 << 16
F_TRACK_PRR = 1 << 17

STEP_MY_STEP = 0
STEP_MY_STEP_CH = 1
STEP_DESIGN = 2

DT_F32 = 0
DT_F64 = 1

OPT_ENV_OFFSET = 1
OPT_KERNEL_PATH = 2
PATH_AUTO = 0
PATH_GENERAL = 1
PATH_LARGE = 2

KERNEL_GENERAL = 0
KERNEL_FAST64 = 1
KERNEL_WIDE = 2
KERNEL_OBSERVE = 3
KERNEL_LARGE = 4
KERNEL_RICH = 16
KERNEL_EXTRA = 32
KERNEL_CH = 64
KERNEL_RING = 128
KERNEL_PACKED = 256
KERNEL_POLICY = 512

MAX_USERS = 4096
MAX_CHANNELS = 4096
MAX_BINS = 1024
SMALL_MAX_USERS = 256        # the one-workgroup kernels; beyond: csrc/step_large.hpp (KERNEL_LARGE)
SMALL_MAX_CHANNELS = 256
SMALL_MAX_BINS = 64

M_SLOTS, M_SUM_REWARD, M_TX_SOLE, M_TX_COLLIDED, M_PRR_SUM, M_PRR_CNT = range(6)
M_COLUMNS = 6

OK = 0
ERR_BAD_ARG = -1
ERR_BAD_CONFIG = -2
ERR_UNSUPPORTED = -3
ERR_HIP = -4
ERR_NO_DEVICE = -5
ERR_ACTION_RANGE = -6
ERR_SEQ_OVERFLOW = -7
ERR_CAPTURE = -8
ERR_TABLE_CONFLICT = -9
ERR_PIGGY_NO_TX = -10


class DiralCfg(ctypes.Structure):
    """ctypes image of ``struct DiralCfg`` (include/diral_env.h)."""

    _fields_ = [
        ("struct_bytes", ctypes.c_uint32),
        ("flags", ctypes.c_uint32),
        ("num_users", ctypes.c_int32),
        ("num_channels", ctypes.c_int32),
        ("num_bins", ctypes.c_int32),
        ("reward_design", ctypes.c_int32),
        ("state_type", ctypes.c_int32),
        ("posdist_type", ctypes.c_int32),
        ("episode_interval", ctypes.c_int32),
        ("info_age_limit", ctypes.c_int32),
        ("pf_threshold", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("pf_penalty", ctypes.c_double),
        ("highway_length", ctypes.c_double),
        ("highway_height", ctypes.c_double),
        ("communication_range", ctypes.c_double),
        ("bin_range", ctypes.c_double),
    ]


class DiralSlotPolicy(ctypes.Structure):
    """ctypes image of ``struct DiralSlotPolicy`` (include/diral_env.h): the policy epilogue of
    ``diral_env_step_policy``."""

    _fields_ = [
        ("struct_bytes", ctypes.c_uint32),
        ("shape_flags", ctypes.c_int32),
        ("pen_threshold", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("pen_value", ctypes.c_double),
        ("shaped_out", ctypes.c_void_p),
        ("sum_r_out", ctypes.c_void_p),
        ("collision_out", ctypes.c_void_p),
        ("pen_counter", ctypes.c_void_p),
        ("pen_prev_actions", ctypes.c_void_p),
        ("sps_prev_action", ctypes.c_void_p),
        ("sps_counter", ctypes.c_void_p),
        ("rssi_threshold", ctypes.c_double),
        ("inc_db", ctypes.c_double),
        ("keep_prob", ctypes.c_double),
        ("draw_counter", ctypes.c_void_p),
        ("draw_keep", ctypes.c_void_p),
        ("draw_choice", ctypes.c_void_p),
        ("seed", ctypes.c_uint64),
        ("seed_clock", ctypes.c_void_p),
        ("actions_out", ctypes.c_void_p),
        ("slots", ctypes.c_int32),
        ("reserved1", ctypes.c_int32),
        ("vel_seed", ctypes.c_uint64),
    ]


class ConfigError(ValueError):
    """A config the reference itself cannot run, or one this build rejects."""


# ---- the reference's State block (test_env.py:26-41) -----------------------

@dataclass
class StateConfig:
    type: int = 2
    add_reward: bool = False
    add_action: bool = True
    add_index: bool = False
    add_velocity: bool = False
    action_index: str = "binary"        # "binary" | "real"  (test_env.py:32)
    piggybacking: bool = False          # test_env.py:33, 71-79, 241-254: needs type 2 + add_channel_obs (validate())
    add_position: bool = False
    add_positional_dist: bool = False
    add_positional_dist_piggy: bool = False
    add_positional_dist_type: int = 2
    num_bins: int = 20
    add_channel_obs: bool = False


@dataclass
class EnvConfig:
    """Mirror of the kwargs of ``TestEnv.__init__`` (test_env.py:12-48).

    Field names and defaults are the reference's.  Keys the hot path never
    reads (``topology``, ``radius``, ``load_file_pos``) are accepted and kept in
    ``extra``.
    """

    num_users: int = 3
    num_channels: int = 3
    mobility: bool = False
    mobility_vary: bool = False
    enable_design_topology: bool = False
    highway_length: float = 200
    enable_fingerprint: bool = False
    reward_design: int = 1
    communication_range: float = 1
    proportional_fair: bool = False
    load_positions: bool = False
    bin_range: float = 500
    congestion_test: bool = False
    State: StateConfig = field(default_factory=StateConfig)
    # driver-level key (main_test.py:22 / :226), needed for `done`
    episode_interval: int = 25
    # build extensions
    track_arrival: bool = False
    track_prr: bool = False
    extra: Dict[str, Any] = field(default_factory=dict)

    # -- constructors ---------------------------------------------------------
    @classmethod
    def from_dict(cls, d: Mapping[str, Any], **overrides: Any) -> "EnvConfig":
        """Accept the reference's ``EnvironmentTest`` dict verbatim."""
        d = dict(d)
        d.update(overrides)
        if "State" not in d or not d["State"]:
            # the reference does `self.state_parameters["type"]` on the default
            # False and raises TypeError (test_env.py:26-27)
            raise ConfigError("EnvironmentTest.State block is required (test_env.py:26-27)")
        state_in = dict(d.pop("State"))
        sfields = {f.name for f in dataclasses.fields(StateConfig)}
        required = ["type", "add_reward", "add_action", "add_index", "add_velocity",
                    "action_index", "piggybacking", "add_position", "add_positional_dist",
                    "add_positional_dist_piggy", "add_positional_dist_type", "num_bins",
                    "add_channel_obs"]
        missing = [k for k in required if k not in state_in]
        if missing:
            # the reference indexes these keys directly -> KeyError (test_env.py:27-41)
            raise ConfigError("State block lacks keys %s (test_env.py:27-41)" % missing)
        state = StateConfig(**{k: v for k, v in state_in.items() if k in sfields})
        efields = {f.name for f in dataclasses.fields(cls)} - {"State", "extra"}
        known = {k: v for k, v in d.items() if k in efields}
        extra = {k: v for k, v in d.items() if k not in efields}
        return cls(State=state, extra=extra, **known)

    @classmethod
    def from_yaml(cls, path: str, **overrides: Any) -> "EnvConfig":
        """Load a reference experiment YAML (configs/4ue_3r_toy/*.yaml): takes the
        ``EnvironmentTest`` block and the top-level ``episode_interval``."""
        import yaml

        with open(path) as fh:
            doc = yaml.safe_load(fh)
        env = dict(doc["EnvironmentTest"]) if "EnvironmentTest" in doc else dict(doc)
        if "episode_interval" in doc and "episode_interval" not in env:
            env["episode_interval"] = doc["episode_interval"]
        return cls.from_dict(env, **overrides)

    # -- derived --------------------------------------------------------------
    @property
    def action_space(self) -> int:          # test_env.py:44
        return self.num_channels

    @property
    def state_space(self) -> int:           # test_env.py:49-85
        s, st = 0, self.State
        if st.add_action:
            if st.action_index == "binary":
                s += self.num_channels
            elif st.action_index == "real":
                s += 1
        if st.add_channel_obs:
            s += self.num_channels
        if st.add_reward:
            s += 1
        if st.add_index:
            s += 1
        if st.add_velocity:
            s += 1
        if st.add_position:
            s += 2
        if st.add_positional_dist:
            s += self.num_users - 1
        if st.piggybacking:                 # test_env.py:71-72
            s += self.num_channels * (self.num_channels - 1)
        if self.enable_fingerprint:
            s += 2
        if st.add_positional_dist_piggy:
            s += st.num_bins
        return s

    @property
    def chobs_width(self) -> int:
        """Length of one agent's `obs` as my_step returns it: A, or A * A when State.piggybacking
        inserts the closest transmitters' previous observations (test_env.py:241-254, 263-264)."""
        return self.num_channels * self.num_channels if self.State.piggybacking else self.num_channels

    def flags(self) -> int:
        st = self.State
        f = 0
        f |= F_MOBILITY if self.mobility else 0
        f |= F_MOBILITY_VARY if self.mobility_vary else 0
        f |= F_TOY_WEIGHTS if self.congestion_test else 0
        f |= F_ADD_ACTION if st.add_action else 0
        f |= F_ACTION_REAL if st.action_index == "real" else 0
        f |= F_ADD_CHANNEL_OBS if st.add_channel_obs else 0
        f |= F_ADD_REWARD if st.add_reward else 0
        f |= F_ADD_INDEX if st.add_index else 0
        f |= F_ADD_VELOCITY if st.add_velocity else 0
        f |= F_ADD_POSITION if st.add_position else 0
        f |= F_ADD_POSDIST if st.add_positional_dist else 0
        f |= F_ADD_POSDIST_PIGGY if st.add_positional_dist_piggy else 0
        f |= F_FINGERPRINT if self.enable_fingerprint else 0
        f |= F_PROPORTIONAL_FAIR if self.proportional_fair else 0
        f |= F_DESIGN_TOPOLOGY if self.enable_design_topology else 0
        f |= F_PIGGYBACKING if st.piggybacking else 0
        f |= F_TRACK_ARRIVAL if self.track_arrival else 0
        f |= F_TRACK_PRR if self.track_prr else 0
        return f

    def validate(self) -> None:
        """Host-side checks shared by every backend (same rules as
        ``diral_env_validate`` in csrc/)."""
        st = self.State
        if self.num_users < 1 or self.num_channels < 1:
            raise ConfigError("num_users and num_channels must be >= 1")
        if st.piggybacking and st.type != 2:
            raise ConfigError("State.piggybacking with State.type 1: the reference inserts only on idle resources there "
                              "(test_env.py:226-232 has no piggybacking branch, :250-254 has), so the length of an "
                              "agent's observation depends on the slot's traffic - no fixed state vector exists")
        if st.piggybacking and not st.add_channel_obs:
            raise ConfigError("State.piggybacking without State.add_channel_obs: get_state_space() counts A*(A-1) columns "
                              "(test_env.py:71-72) that obtain_state never writes (test_env.py:539-541)")
        if st.action_index not in ("binary", "real"):
            raise ConfigError("action_index must be 'binary' or 'real' (test_env.py:50-55)")
        if st.type not in (1, 2):
            raise ConfigError("State.type must be 1 or 2 (test_env.py:226-240)")
        if self.reward_design not in (1, 2, 3, 4, 5):
            raise ConfigError("reward_design must be 1..5 (test_env.py:170-199; "
                              "anything else is undefined behaviour in the reference)")
        if st.add_positional_dist_piggy and st.add_positional_dist_type not in (1, 2):
            raise ConfigError("add_positional_dist_type must be 1 or 2 (test_env.py:555-560)")
        if not (self.mobility or self.enable_design_topology):
            raise ConfigError("the reference builds vehicles only with mobility or "
                              "enable_design_topology (network.py:54-60)")
        if st.add_positional_dist_piggy and st.num_bins < 1:
            raise ConfigError("num_bins must be >= 1")
        if self.episode_interval < 1:
            raise ConfigError("episode_interval must be >= 1")

    def to_c(self) -> DiralCfg:
        self.validate()
        c = DiralCfg()
        c.struct_bytes = ctypes.sizeof(DiralCfg)
        c.flags = self.flags()
        c.num_users = int(self.num_users)
        c.num_channels = int(self.num_channels)
        c.num_bins = int(self.State.num_bins)
        c.reward_design = int(self.reward_design)
        c.state_type = int(self.State.type)
        c.posdist_type = int(self.State.add_positional_dist_type)
        c.episode_interval = int(self.episode_interval)
        c.info_age_limit = 20               # network.py:547
        c.pf_threshold = 10                 # test_env.py:89
        c.pf_penalty = -10.0                # test_env.py:90
        c.highway_length = float(self.highway_length)
        c.highway_height = 2.0              # network.py:31
        c.communication_range = float(self.communication_range)
        c.bin_range = float(self.bin_range)
        return c

    def replace(self, **kw: Any) -> "EnvConfig":
        state_kw = kw.pop("State", None)
        new = dataclasses.replace(self, **kw)
        if state_kw:
            new.State = dataclasses.replace(self.State, **state_kw)
        else:
            new.State = dataclasses.replace(self.State)
        return new


def c2_config(**kw: Any) -> EnvConfig:
    """BASELINE.json configs[1]: 64 UE / 32 resources;
    every flag as in the toy YAML except congestion_test."""
    return bench_config(64, 32, 2000.0, **kw)


def bench_config(n: int, a: int, length: float, **kw: Any) -> EnvConfig:
    cfg = EnvConfig(
        num_users=n, num_channels=a, mobility=True, highway_length=length,
        reward_design=2, communication_range=250, bin_range=500,
        congestion_test=False,
        State=StateConfig(type=2, add_action=True, action_index="binary",
                          add_positional_dist_piggy=True, add_positional_dist_type=2,
                          num_bins=20))
    return cfg.replace(**kw) if kw else cfg
