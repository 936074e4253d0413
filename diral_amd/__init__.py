"""diral_amd - MI355X-native batched V2V resource-allocation environment.

The one data-parallel hot path of gundoganalperen/DIRAL (envs/test_env.py +
network.py + vehicle.py: the env step) as hand-written HIP for gfx950 behind a
C-ABI (include/diral_env.h), with a host-side mirror of the reference's env
interface.  See DESIGN.md.
"""
from .config import (ConfigError, EnvConfig, StateConfig, STEP_DESIGN, STEP_MY_STEP, STEP_MY_STEP_CH,
                     bench_config, c2_config)

__all__ = ["EnvConfig", "StateConfig", "ConfigError", "VecV2VEnv", "StreamedVecEnv", "TestEnv", "bench_config", "c2_config",
           "STEP_MY_STEP", "STEP_MY_STEP_CH", "STEP_DESIGN"]


def __getattr__(name):
    # torch-dependent classes are imported lazily so config/oracle tooling can
    # be used without touching the GPU stack
    if name == "VecV2VEnv":
        from .vec_env import VecV2VEnv
        return VecV2VEnv
    if name == "StreamedVecEnv":
        from .streamed import StreamedVecEnv
        return StreamedVecEnv
    if name == "TestEnv":
        from .compat import TestEnv
        return TestEnv
    raise AttributeError(name)
