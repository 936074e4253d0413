"""Roofline bookkeeping for the fused step kernel.

Algorithmic bytes per env-slot use the CANONICAL reference state, not this
build's packed layout: a table entry is 16 B (x f64, seq i32, age i32; y is
derivable), read once and written once per slot, plus per-vehicle state and the
outputs:

    bytes = 2*16*N^2 + N*(4 + 16 + 8 + 8) + 4*N*A + 4*N + 4*N*S

(actions i32, pos_x r/w, pos_y, vel; channel-obs f32; reward f32; state f32).
C2 (N=64, A=32, S=52): 155136 B/env-slot = 2424 B/agent-step.  This is the figure
`roofline.achieved` is built from (the task's definition).

The packed layout of this build moves 12 B per entry (u32 key + f64 x), so the
bytes that really cross the HBM interface are fewer: `layout_bytes_per_env_slot`
(matches the rocprofv3 FETCH_SIZE / WRITE_SIZE counters within a few percent,
profiles/README.md).  bench.py reports both; the kernels are VALU-bound, and the
layout figure over the kernel time is their real HBM rate.
"""
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable


def algorithmic_bytes_per_env_slot(n: int, a: int, s: int) -> int:
    return 2 * 16 * n * n + n * (4 + 16 + 8 + 8) + 4 * n * a + 4 * n + 4 * n * s


def layout_bytes_per_env_slot(n: int, a: int, s: int, emit_chobs: bool, out_bytes: int = 4) -> int:
    """What csrc/step_fast64.hpp / step_wide.hpp move per env-slot: every table word read and
    written once at 12 B per entry, the per-vehicle arrays, reward and state, and the channel
    observation only when it is requested."""
    return 2 * 12 * n * n + n * (4 + 16 + 8 + 8) + out_bytes * n + out_bytes * n * s + (out_bytes * n * a if emit_chobs else 0)
