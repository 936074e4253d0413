"""Roofline bookkeeping for the fused step kernel.

Algorithmic bytes per env-slot use the CANONICAL reference state, not this
build's packed layout: a table entry is 16 B (x f64, seq i32, age i32; y is
derivable), read once and written once per slot, plus per-vehicle state and the
outputs:

    bytes = 2*16*N^2 + N*(4 + 16 + 8 + 8) + 4*N*A + 4*N + 4*N*S

(actions i32, pos_x r/w, pos_y, vel; channel-obs f32; reward f32; state f32).
C2 (N=64, A=32, S=52): 155136 B/env-slot = 2424 B/agent-step.
The packed layout actually moves 12 B per entry, so measured HBM traffic is
BELOW this figure.
"""
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable


def algorithmic_bytes_per_env_slot(n: int, a: int, s: int) -> int:
    return 2 * 16 * n * n + n * (4 + 16 + 8 + 8) + 4 * n * a + 4 * n + 4 * n * s


def actual_table_bytes_per_env_slot(n: int) -> int:
    """what csrc/step_kernel.hpp moves for the table: u32 key + f64 x, r+w"""
    return 2 * 12 * n * n
