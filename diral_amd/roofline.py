"""Roofline bookkeeping for the fused step kernel.

Algorithmic bytes per env-slot use the CANONICAL reference state, not this
build's packed layout: a table entry is 16 B (x f64, seq i32, age i32; y is
derivable), read once and written once per slot, plus per-vehicle state and the
outputs:

    bytes = 2*16*N^2 + N*(4 + 16 + 8 + 8) + 4*N*A + 4*N + 4*N*S

(actions i32, pos_x r/w, pos_y, vel; channel-obs f32; reward f32; state f32).
C2 (N=64, A=32, S=52): 155136 B/env-slot = 2424 B/agent-step.  This is the figure
`roofline.achieved` is built from (the task's definition).

The layout of this build moves fewer bytes: per table entry one byte of thermometer code (the
entry's lag behind its subject) and one byte of age (a 4-byte (seq, age) word for step_wide's
plane form: sparse topologies at N > 64) - the xpos of an entry that lags its subject by at most 7 stamps comes
from an 8-deep per-subject ring (csrc/step_fast64.hpp, step_wide.hpp), and in
steady state that is every entry; the per-entry planes are touched only for older entries.
`layout_bytes_per_env_slot` is that figure (the rocprofv3 FETCH_SIZE / WRITE_SIZE counters show
it plus the register-spill scratch of the kernel, profiles/README.md).  bench.py reports both;
the layout figure over the kernel time is the real HBM rate.
"""
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
HBM_ACHIEVABLE_GBPS = 6290.0   # measured float4 copy (MI355X_MICROARCH.md: 79 % of the spec)
INFINITY_CACHE_BYTES = 256 << 20   # die-level L3 in front of HBM (MI355X_MICROARCH.md)


def algorithmic_bytes_per_env_slot(n: int, a: int, s: int) -> int:
    return 2 * 16 * n * n + n * (4 + 16 + 8 + 8) + 4 * n * a + 4 * n + 4 * n * s


def packed_table(n: int, communication_range: float, highway_length: float) -> bool:
    """The table form csrc/diral_env.hip's use_packed_table chooses for a handle (without DIRAL_TABLE_FORM): packed
    codes + ages (2 B per entry) for step_fast64 (N <= 64) and for step_wide where on average at least 15 (N <= 128) /
    20 vehicles sit within communication range (BASELINE configs[2] and [4]); a sparser highway at N > 64 keeps the
    4-byte (seq, age) word.  Callers that have a handle pass `packed = bool(env.last_kernel() & KERNEL_PACKED)` instead."""
    if n <= 64:
        return True
    neigh = n * 2.0 * communication_range / highway_length
    return neigh >= (15.0 if n <= 128 else 20.0)


def layout_bytes_per_env_slot(n: int, a: int, s: int, emit_chobs: bool, out_bytes: int = 4, *, packed: bool) -> int:
    """What csrc/step_fast64.hpp / step_wide.hpp HAVE to move per env-slot in steady state - the compulsory
    bytes of this build's table layout: every table entry's stored form read and written once (packed: one
    code byte + one age byte; else the 4-byte (seq, age) word), the subjects' ring rows read and one stamp each
    written, the subjects' own sequence numbers (packed form), the per-vehicle arrays, reward and state, and
    the channel observation only when it is requested."""
    entry = 2 if packed else 4
    table = 2 * entry * n * n + 64 * n + 8 * n + (8 * n if packed else 0)
    return table + n * (4 + 16 + 8 + 8) + out_bytes * n + out_bytes * n * s + (out_bytes * n * a if emit_chobs else 0)


def resident_bytes_per_env(n: int, *, packed: bool) -> int:
    """The state a launch RE-READS from the launch before - stored table words (read and written in place), ring rows, own
    sequence numbers, per-vehicle arrays: what must survive in a cache between two launches for the reads of the second
    not to reach DRAM.  The outputs are written once and never read back by the kernels."""
    entry = 2 if packed else 4
    return entry * n * n + 64 * n + (4 * n if packed else 0) + n * (4 + 8 + 8 + 8)


def memory_level(n: int, a: int, s: int, batch: int, emit_chobs: bool, out_bytes: int = 4, *, packed: bool) -> dict:
    """Where the bytes of one launch can come from / go to: `resident` = batch x resident_bytes_per_env against the 256 MiB
    Infinity Cache.  The rocprofv3 FETCH_SIZE / WRITE_SIZE counters (and the layout bytes) count requests at the L2 <->
    fabric interface: when the resident state fits the Infinity Cache they are FABRIC bytes - an upper bound on DRAM
    traffic - and only the outputs (streamed, never read back) certainly reach HBM."""
    resident = batch * resident_bytes_per_env(n, packed=packed)
    outputs = batch * (out_bytes * n + out_bytes * n * s + (out_bytes * n * a if emit_chobs else 0))
    fits = resident <= INFINITY_CACHE_BYTES
    return {"resident_bytes": resident, "output_bytes": outputs, "infinity_cache_bytes": INFINITY_CACHE_BYTES,
            "reads_served_by": "infinity cache (resident state fits: fabric bytes, an upper bound on DRAM reads)" if fits
                               else "hbm (resident state exceeds the Infinity Cache)"}
