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


def packed_table(n: int) -> bool:
    """The DEFAULT table form of the BASELINE configurations: packed codes + ages (2 B per entry) for step_fast64
    (N <= 64) and for step_wide on dense topologies (configs[2] and [4]); a sparse highway at N > 64 keeps the 4-byte
    (seq, age) word (csrc/step_wide.hpp).  The runtime chooses per handle (density, DIRAL_TABLE_FORM: csrc/diral_env.hip
    use_packed_table): callers that have a handle pass `packed = bool(env.last_kernel() & KERNEL_PACKED)` to
    layout_bytes_per_env_slot instead of relying on this."""
    return True


def layout_bytes_per_env_slot(n: int, a: int, s: int, emit_chobs: bool, out_bytes: int = 4, packed=None) -> int:
    """What csrc/step_fast64.hpp / step_wide.hpp HAVE to move per env-slot in steady state - the compulsory
    bytes of this build's table layout: every table entry's stored form read and written once (packed: one
    code byte + one age byte; else the 4-byte (seq, age) word), the subjects' ring rows read and one stamp each
    written, the subjects' own sequence numbers (packed form), the per-vehicle arrays, reward and state, and
    the channel observation only when it is requested."""
    if packed is None:
        packed = packed_table(n)
    entry = 2 if packed else 4
    table = 2 * entry * n * n + 64 * n + 8 * n + (8 * n if packed else 0)
    return table + n * (4 + 16 + 8 + 8) + out_bytes * n + out_bytes * n * s + (out_bytes * n * a if emit_chobs else 0)


def resident_bytes_per_env(n: int, packed=None) -> int:
    """The state a launch RE-READS from the launch before - stored table words (read and written in place), ring rows, own
    sequence numbers, per-vehicle arrays: what must survive in a cache between two launches for the reads of the second
    not to reach DRAM.  The outputs are written once and never read back by the kernels."""
    if packed is None:
        packed = packed_table(n)
    entry = 2 if packed else 4
    return entry * n * n + 64 * n + (4 * n if packed else 0) + n * (4 + 8 + 8 + 8)


def memory_level(n: int, a: int, s: int, batch: int, emit_chobs: bool, out_bytes: int = 4, packed=None) -> dict:
    """Where the bytes of one launch can come from / go to: `resident` = batch x resident_bytes_per_env against the 256 MiB
    Infinity Cache.  The rocprofv3 FETCH_SIZE / WRITE_SIZE counters (and the layout bytes) count requests at the L2 <->
    fabric interface: when the resident state fits the Infinity Cache they are FABRIC bytes - an upper bound on DRAM
    traffic - and only the outputs (streamed, never read back) certainly reach HBM."""
    resident = batch * resident_bytes_per_env(n, packed)
    outputs = batch * (out_bytes * n + out_bytes * n * s + (out_bytes * n * a if emit_chobs else 0))
    fits = resident <= INFINITY_CACHE_BYTES
    return {"resident_bytes": resident, "output_bytes": outputs, "infinity_cache_bytes": INFINITY_CACHE_BYTES,
            "reads_served_by": "infinity cache (resident state fits: fabric bytes, an upper bound on DRAM reads)" if fits
                               else "hbm (resident state exceeds the Infinity Cache)"}
