#!/usr/bin/env python3
"""bench.py - headline benchmark: agent-steps/sec of the fused V2V env step.

Metric (BASELINE.json): agent-steps/sec (envs x vehicles) on the 64-UE/32-res
batched env.  One "step" = one time-slot of ALL B envs of this rank = one launch
of the fused kernel (stamp -> collide -> reward -> closest-tx -> gossip merge ->
move -> observe), actions pre-generated in HBM, outputs written to HBM.

  python bench.py --gpus 1 --steps 200 --warmup 50
  python bench.py --gpus N ...          (N > 1 without a launcher: re-executes itself as N ranks under
                                         torch.distributed.run on 127.0.0.1 with a free port, diral_amd/spawn.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; envs are independent so ranks share nothing on the step
path (weak scaling: B per GPU fixed); RCCL is used once, to all-reduce the
episode metrics.  Rank 0 prints ONE JSON line.

Timed region: after the reset every run first plays untimed pre-roll slots (at least
PREROLL = 60, SURVEY 8d: >= 50, past the ghost-entry / table-filling phase of the first
19 slots, Q4; and at least PREROLL_SECONDS, so that a GPU coming out of idle has reached
its clocks), then the W warm-up slots it was asked for, then exactly K timed slots between
barrier + synchronize pairs.  The kernel's own duration is measured with HIP events
on the launch stream over the same K launches.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from diral_amd.config import (KERNEL_CH, KERNEL_EXTRA, KERNEL_FAST64, KERNEL_LARGE, KERNEL_PACKED, KERNEL_RICH, KERNEL_WIDE,  # noqa: E402
                              bench_config)
from diral_amd.metrics import gather_metrics  # noqa: E402
from diral_amd.roofline import (HBM_ACHIEVABLE_GBPS, HBM_PEAK_GBPS, algorithmic_bytes_per_env_slot, memory_level,  # noqa: E402
                                layout_bytes_per_env_slot)
from diral_amd.spawn import check_visible_gpus, spawn_ranks, under_launcher  # noqa: E402
from diral_amd.vec_env import VecV2VEnv  # noqa: E402

WORKLOADS = {
    # name: (N, A, L, B per GPU, mobility_vary)  - SURVEY.md section 8 table
    "c2": (64, 32, 2000.0, 4096, False),    # BASELINE.json configs[1]: the metric's config
    "c3": (256, 64, 4000.0, 8192, False),   # configs[2] congested
    "c5": (128, 64, 4000.0, 16384, True),   # configs[4] dynamic density
    "c4shard": (64, 32, 2000.0, 32768, False),   # configs[3]: the per-GPU share of 262144 envs over 8 GPUs
    # beyond every BASELINE.json configuration: a size only the three-launch form runs (csrc/step_large.hpp; same density as c3)
    "n1024": (1024, 64, 16000.0, 256, False),
}
PREROLL = 60      # untimed slots after every reset, before the requested warm-up (>= 50)
PREROLL_SECONDS = 0.3   # ... and at least this long: a GPU coming out of idle needs tens of ms to reach its clocks
GLOBAL_SEED = 1234


def usable_cores() -> int:
    """Host threads this process may really use: sched affinity capped by the
    cgroup CPU quota (a 256-thread box with a 16-CPU quota has 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _time_oracle(cfg, threads: int, seconds_target: float, native: bool = False):
    import numpy as np
    from oracle.oracle import Oracle, SQ_POW
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    B = max(threads * 16, 16)
    rng = np.random.default_rng(GLOBAL_SEED)
    o = Oracle(cfg, batch=B, sq_mode=SQ_POW, threads=threads, native=native)
    o.reset(rng.integers(0, int(L), size=(B, N)).astype(np.float64), np.zeros((B, N)),
            rng.uniform(1.1, 2.7, size=(B, N)))
    warm, slots = 25, 0
    acts = rng.integers(0, A, size=(64, B, N)).astype(np.int32)
    for t in range(warm):
        r, c = o.step(0, acts[t % 64], t)
        o.obtain_state(acts[t % 64], c, r)
    t0 = time.perf_counter()
    t = warm
    while True:
        r, c = o.step(0, acts[t % 64], t)
        o.obtain_state(acts[t % 64], c, r)
        t += 1
        slots += 1
        el = time.perf_counter() - t0
        if el >= seconds_target or slots >= 20000:
            break
    return B * N * slots / el, B, slots, warm, el


def _native_oracle_agrees(cfg) -> bool:
    """The -O3 -march=native build of oracle/diral_oracle.c (compiled here, on the box it runs on) against the checker
    build (-O2, generic): rewards, channel observation and state vectors of a short seeded run, bit for bit."""
    import numpy as np
    from oracle.oracle import Oracle, SQ_POW
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    runs = []
    for native in (False, True):
        rng = np.random.default_rng(77)
        o = Oracle(cfg, batch=4, sq_mode=SQ_POW, threads=2, native=native)
        o.reset(rng.integers(0, int(L), size=(4, N)).astype(np.float64), np.zeros((4, N)), rng.uniform(1.1, 2.7, size=(4, N)))
        out = []
        for t in range(30):
            a = rng.integers(0, A, size=(4, N)).astype(np.int32)
            r, c = o.step(0, a, t)
            out += [r.copy(), c.copy(), o.obtain_state(a, c, r).copy()]
        runs.append(out)
    return all(np.array_equal(x, y) for x, y in zip(*runs))


def cpu_baseline(cfg):
    """Time the CPU restatement (oracle/diral_oracle.c, reference-faithful pow() mode) on this host: at ONE thread and
    on every usable core, on a bounded sample of the same workload.  BASELINE.md section 4 asks for an
    `-O3 -march=native` build: that build is compiled HERE (oracle/Makefile: libdiral_oracle_native.so), held to the
    checker build (-O2, generic: what tests/ use) bit for bit on a short run, and is what gets timed; the checker
    build is timed beside it (`checker_build`).  A reported baseline, not the target."""
    from oracle.oracle import CHECKER_FLAGS, NATIVE_FLAGS, has_openmp
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    cores = usable_cores()
    threads = min(cores, 64) if has_openmp() else 1
    native, why = True, ""
    try:
        if not _native_oracle_agrees(cfg):
            native, why = False, "the native build disagreed with the checker build"
    except Exception as exc:            # no compiler on this box: time the checker build and say so
        native, why = False, "native build failed (%s)" % type(exc).__name__
    v1, b1, s1, warm, e1 = _time_oracle(cfg, 1, 6.0, native)
    vn, bn, sn, _, en = _time_oracle(cfg, threads, 10.0, native) if threads > 1 else (v1, b1, s1, warm, e1)
    chk = None
    if native:
        c1 = _time_oracle(cfg, 1, 3.0, False)
        cn = _time_oracle(cfg, threads, 4.0, False) if threads > 1 else c1
        chk = {"flags": "gcc " + CHECKER_FLAGS, "value": cn[0], "value_1thread": c1[0]}
    N, A = cfg.num_users, cfg.num_channels
    flags = "gcc " + (NATIVE_FLAGS if native else CHECKER_FLAGS)
    return {
        "value": vn, "unit": "agent-steps/s", "cores": threads, "kind": "port",
        "value_1thread": v1, "cpu_model": cpu_model(), "nproc": os.cpu_count(), "usable_cores": cores,
        "build_flags": flags, "checker_build": chk,
        "sample": "the same %d-UE/%d-res workload after %d warm-up slots, oracle/diral_oracle.c (C restatement "
                  "pinned to the reference goldens) built with `%s`%s, my_step + obtain_state per slot: %d envs x %d slots on %d "
                  "OpenMP threads (%.1f s); %d envs x %d slots on 1 thread (%.1f s)" % (
                      N, A, warm, flags, (" [" + why + "]") if why else " on this box and checked against the -O2 checker build",
                      bn, sn, threads, en, b1, s1, e1),
    }


def prr_parity(cfg, device, envs: int = 64, slots: int = 60):
    """PRR parity on a bounded sample (BASELINE.json: 'PRR matching the reference
    within 1e-6'): the same seeded envs and actions through the HIP path
    (my_step_ch, test_env.py:384-405 ratio R per transmission) and through the CPU
    oracle in its reference-faithful mode; episode PRR = mean R over transmissions."""
    import numpy as np
    from oracle.oracle import Oracle, SQ_POW
    from diral_amd.config import STEP_MY_STEP_CH
    N, A, L = cfg.num_users, cfg.num_channels, cfg.highway_length
    rng = np.random.default_rng(4321)
    x0 = rng.integers(0, int(L), size=(envs, N)).astype(np.float64)
    v0 = rng.uniform(1.1, 2.7, size=(envs, N))
    env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float64, step_mode="my_step_ch")
    env.reset_topology(x0, None, v0)
    orc = Oracle(cfg, batch=envs, sq_mode=SQ_POW, threads=min(usable_cores(), 16))
    orc.reset(x0, np.zeros((envs, N)), v0)
    for t in range(slots):
        a = rng.integers(0, A, size=(envs, N)).astype(np.int32)
        env.step(a, t)
        orc.step(STEP_MY_STEP_CH, a, t)
    m, om = env.metrics().cpu().numpy(), orc.metrics()
    gpu = float(m[:, 4].sum() / m[:, 5].sum())
    cpu = float(om[:, 4].sum() / om[:, 5].sum())
    per_env = np.abs(m[:, 4] / m[:, 5] - om[:, 4] / om[:, 5]).max()
    return {"gpu": gpu, "cpu_oracle": cpu, "abs_diff": abs(gpu - cpu), "max_abs_diff_per_env": float(per_env),
            "tolerance": 1e-6, "sample": "%d envs x %d slots, my_step_ch, reward_design %d" % (envs, slots, cfg.reward_design)}


def kernel_name(code: int, N: int, out_dtype: str) -> str:
    """Name of the instantiation `diral_env_last_kernel` reports (what rocprofv3's
    kernel trace shows); all y == 0 in the bench topologies."""
    def b(x):
        return "true" if x else "false"
    fam = code & 15
    if fam == KERNEL_LARGE:
        return "diral::large_search_kernel + diral::large_mergen_kernel<%s> + diral::large_hist_kernel" % (
            "4,4" if N <= 256 else "8,2" if N <= 512 else "16,2") if N <= 1024 else "diral::large_search_kernel + diral::large_merge_kernel + diral::large_hist_kernel"
    o64, ch, extra, rich = out_dtype == "f64", bool(code & KERNEL_CH), bool(code & KERNEL_EXTRA), bool(code & KERNEL_RICH)
    if fam == KERNEL_FAST64:
        return "diral::step_fast64_kernel<true,%s,%s,%s,%s>" % (b(o64), b(ch), b(extra), b(rich))
    if fam == KERNEL_WIDE:
        return "diral::step_wide_kernel<%d,%s,%s,%s,%s,%s,%s>" % (2 if N <= 128 else 4, b(o64), b(N in (128, 256)), b(ch),
                                                                b(extra), b(rich), b(code & KERNEL_PACKED))
    return "diral::step_kernel<%d,%s>" % (1 if N <= 64 else 2 if N <= 128 else 4, b(out_dtype == "f32" and not ch))


def load_pmc(workload: str):
    """The committed PMC record of this bench command (profiles/pmc_counters.json, written by
    profiles/make_pmc_json.py from the rocprofv3 summaries of separate --pmc passes): HBM bytes per
    launch, VALU busy fraction and the shader clock measured in the same passes.  NOT measured in
    this run - rocprofv3 cannot attach to the process it did not start."""
    for fn in ("pmc_counters.json", "pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as fh:
                rec = json.load(fh).get(workload)
            if rec:
                return rec
        except Exception:
            pass
    return None


def csrc_digest() -> str:
    """Digest of the kernel sources the library is built from (diral_amd/csrc, include/diral_env.h): stamped into every
    profile summary (profiles/run_profile.sh) and into profiles/pmc_counters.json, so that a bench line can say whether
    the committed counters describe the kernels it ran (`roofline.pmc_current`) - the GPU boxes have no .git."""
    from diral_amd.build import source_digest
    return source_digest()


def issue_roofline(pmc, kernel_ms):
    pw = (pmc or {}).get("per_wave") or {}
    if not pw.get("valu") or not pw.get("waves") or not pmc.get("clock_GHz"):
        return None
    insts = pw["valu"] * pw["waves"]
    need = insts * 4.0
    have = 256 * 4 * pmc["clock_GHz"] * 1e9 * kernel_ms * 1e-3
    return {"pipe": "valu", "valu_insts_per_launch": insts, "cycles_per_inst": 4, "simds": 1024,
            "needed_simd_cycles": need, "available_simd_cycles": have, "frac": need / have,
            "floor_ms": kernel_ms * need / have,
            "salu_insts_per_launch": pw.get("salu", 0) * pw["waves"], "lds_insts_per_launch": pw.get("lds", 0) * pw["waves"],
            "note": "instruction counts from the committed rocprofv3 --pmc passes (pmc_current says whether they describe the "
                    "kernels this run timed); kernel time of THIS run"}


def roofline_object(res, pmc):
    """The `roofline` object of a measurement: what bounds the kernel and how far it is from it.

    `achieved` / `frac`: the bytes this build's exact minimal state layout HAS to move per launch
    (diral_amd/roofline.py layout_bytes_per_env_slot: every 4-byte table key read and written once, the
    subjects' xpos rings, per-vehicle arrays, the outputs) over the kernel time measured in this run, over
    the 8 TB/s HBM peak - a fraction of the HBM roofline, <= 1 by construction.  `counter_frac`: the same
    with the HBM bytes the PMC counters saw (committed passes).  `bound` names the busiest pipe: `valu_busy` /
    `lds_busy` are busy cycles over (shader clock x kernel duration), the clock from GRBM_GUI_ACTIVE of the same
    session; `cu_busy` is the share of the kernel's duration a CU holds at least one wave (the 2.29-round tail of
    4096 workgroups on 256 x 7 slots shows here).
    SURVEY 8d's canonical byte model (16-byte table entry) is reported as `model_*`: a throughput in the
    model's units, not an HBM utilisation (this layout does not move two thirds of those bytes)."""
    k_s = res["kernel_ms"] * 1e-3
    traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
    pmc_ms = (pmc.get("kernel_ms_traffic_passes") or pmc.get("kernel_ms_profiled")) if pmc else None
    counter_frac = (traffic / ((pmc_ms or res["kernel_ms"]) * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None
    # what limits: the busiest of the three pipes the committed counters cover (VALU, LDS, HBM)
    pipes = {"valu": (pmc or {}).get("valu_busy") or 0.0, "lds": (pmc or {}).get("lds_busy") or 0.0,
             "hbm": counter_frac or res["layout_rate_GBps"] / HBM_PEAK_GBPS}
    out = {
        "bound": max(pipes, key=pipes.get),
        "achieved": res["layout_rate_GBps"],
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": res["layout_rate_GBps"] / HBM_PEAK_GBPS,
        "bytes_per_launch": res["layout_bytes_per_launch"],
        "bytes_note": "compulsory HBM bytes of this build's table layout per launch (per entry one code byte + one age byte read "
                      "+ written - a 4-byte (seq, age) word for step_wide's plane form on sparse topologies -, per-subject xpos rings and sequence numbers, "
                      "per-vehicle arrays, outputs): diral_amd/roofline.py",
        "kernel": res["kernel"],
        "kernel_ms": res["kernel_ms"],
        "traffic": traffic,
        "counter_frac": counter_frac,
        "valu_frac": pmc.get("valu_busy") if pmc else None,      # the fraction of the pipe that bounds the kernel (= valu_busy)
        "valu_busy": pmc.get("valu_busy") if pmc else None,
        "lds_busy": pmc.get("lds_busy") if pmc else None,
        "cu_busy": pmc.get("cu_busy") if pmc else None,
        "clock_GHz": pmc.get("clock_GHz") if pmc else None,
        # the roofline of the pipe that bounds the kernel (`bound`): the VALU cycles its instruction mix NEEDS - vector
        # instructions per launch (SQ_INSTS_VALU of the committed passes) x 4 cycles each (a 64-lane wave on a 16-lane SIMD;
        # the float64 instructions of this path issue at that rate on gfx950 too) - over the SIMD-cycles the chip HAS
        # during the kernel (256 CUs x 4 SIMDs x shader clock x kernel time of this run).  1.0 = no launch with this
        # instruction count can be faster; floor_ms = the kernel time at which it would be 1.0
        "issue": issue_roofline(pmc, res["kernel_ms"]),
        "pmc_source": pmc.get("source") if pmc else None,
        "pmc_commit": pmc.get("commit") if pmc else None,
        "pmc_csrc_sha": pmc.get("csrc_sha") if pmc else None,
        "csrc_sha": csrc_digest(),
        "pmc_current": (pmc.get("csrc_sha") == csrc_digest()) if pmc and pmc.get("csrc_sha") else None,
        "pmc_note": "traffic / valu_busy / clock_GHz come from the committed rocprofv3 --pmc passes of this command "
                    "(separate runs; counter_frac uses the kernel time of those passes), not from this run" if pmc else None,
        # frac is a fraction of the 8 TB/s SPEC; the copy rate the guide measured is 6.29 TB/s
        "frac_of_achievable": res["layout_rate_GBps"] / HBM_ACHIEVABLE_GBPS,
        # HBM or Infinity Cache?  No gfx950 counter tells a MALL hit from a DRAM access (rocprofv3 --list-avail:
        # TCC_EA0_RDREQ_DRAM counts requests "destined for DRAM (MC)", the memory-side cache sits behind that interface),
        # so the line says what CAN be cache-resident: `memory.reads_served_by`, and DRAM fraction bounds - the outputs
        # alone (streamed, never read back: they reach HBM) ... every layout byte
        "memory": res.get("memory"),
        "dram_frac_bounds": ([res["memory"]["output_bytes"] / k_s / 1e9 / HBM_PEAK_GBPS, res["layout_rate_GBps"] / HBM_PEAK_GBPS]
                             if res.get("memory") and res["memory"]["resident_bytes"] <= res["memory"]["infinity_cache_bytes"]
                             else [res["layout_rate_GBps"] / HBM_PEAK_GBPS] * 2) if res.get("memory") else None,
        "model_bytes_per_launch": res["algorithmic_bytes_per_launch"],
        "model_throughput_TBps": res["algorithmic_bytes_per_launch"] / k_s / 1e12,
        # SURVEY 8d's figure as the contract words it (algorithmic bytes per unit x units per launch / launch time / peak):
        # a BYTE-MODEL ratio, not physical - `frac` / `counter_frac` are the bytes that move
        "model_frac": res["algorithmic_bytes_per_launch"] / k_s / 1e9 / HBM_PEAK_GBPS,
        "model_frac_note": "byte model (SURVEY 8d: 16-byte table entries), not physical: > 1 because the layout stores an entry "
                           "in 2 bytes; frac / counter_frac are physical",
        "model_note": "SURVEY 8d's canonical byte model (16-byte table entry read + written) over the kernel time: a throughput "
                      "in the MODEL's units - this layout moves 2 bytes per entry, so the figure can exceed the 8 TB/s "
                      "peak without the HBM being near it; not a roofline fraction",
    }
    return out


def run_workload(name, device, rank, world, steps, warmup, batch=0, out_dtype="f32", step_mode="my_step",
                 emit_chobs=False, sticky=0.0, use_dist=False):
    """One timed run of one workload on this rank.  Returns (env, result dict)."""
    N, A, L, B, vary = WORKLOADS[name]
    if batch > 0:
        B = batch
    cfg = bench_config(N, A, L, mobility_vary=vary)
    dt = torch.float32 if out_dtype == "f32" else torch.float64
    # one GLOBAL seed; rank r holds envs [r*B, (r+1)*B) of the whole batch (DIRAL_OPT_ENV_OFFSET):
    # the sharded job draws what one GPU holding all world*B envs would draw
    env = VecV2VEnv(cfg, batch=B, device=device, out_dtype=dt, step_mode=step_mode, env_offset=rank * B)
    env.reset_topology(seed=GLOBAL_SEED)

    # synthetic actions, resident in HBM before the timed region: a ring of
    # pre-drawn [B,N] tensors (iid uniform, or sticky to mimic a converged policy)
    ring = 32
    acts = [env.sample(seed=1000 + i) for i in range(ring)]
    if sticky > 0:
        g = torch.Generator(device=device).manual_seed(99 + rank)
        for i in range(1, ring):
            keep = torch.rand((B, N), device=device, generator=g) < sticky
            acts[i] = torch.where(keep, acts[i - 1], acts[i])
    mode = env.step_mode
    ei = cfg.episode_interval

    def one_step(t):
        if emit_chobs:
            env._step(mode, acts[t % ring], t, want_chobs=True)      # state + reward + channel observation
        else:
            env.step(acts[t % ring], t)
        if t % ei == ei - 1:
            env.update_velocity(seed=t)      # main_test.py:226-233 (no-op unless mobility_vary)

    t = 0
    t_pre = time.perf_counter()
    while t < PREROLL or time.perf_counter() - t_pre < PREROLL_SECONDS:
        for _ in range(20):
            one_step(t)
            t += 1
        torch.cuda.synchronize(device)
    preroll = t
    for _ in range(warmup):
        one_step(t)
        t += 1
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize(device)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    # (a torch event creates its HIP event at the first record - 20 us that belong to no step: both exist before the clock starts)
    ev0.record()
    ev1.record()
    torch.cuda.synchronize(device)
    t_start = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        one_step(t)
        t += 1
    ev1.record()
    # (the end of the timed region: the event is polled first - hipEventQuery, a few microseconds of latency - and the
    # blocking synchronize the contract asks for then returns at once; waiting in hipDeviceSynchronize alone adds its
    # interrupt wake-up, ~50 us, to a region that is only K x 70 us long at the driver's K = 20)
    while not ev1.query():
        pass
    torch.cuda.synchronize(device)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize(device)
    wall = time.perf_counter() - t_start
    wall_local = wall
    kernel_ms = ev0.elapsed_time(ev1) / steps       # HIP events on the launch stream
    env.check()
    if use_dist:
        w = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    bytes_launch = algorithmic_bytes_per_env_slot(N, A, cfg.state_space) * B
    code = env.last_kernel()
    from diral_amd.config import KERNEL_PACKED
    layout_launch = layout_bytes_per_env_slot(N, A, cfg.state_space, emit_chobs, packed=bool(code & KERNEL_PACKED)) * B
    res = {
        "workload": name, "N": N, "A": A, "B": B, "L": L, "state_space": cfg.state_space, "cfg": cfg,
        "wall": wall, "ms_per_step": wall / steps * 1e3, "kernel_ms": kernel_ms,
        "agent_steps_per_s": float(B) * N * steps * world / wall,
        "kernel": kernel_name(code, N, out_dtype), "kernel_code": code,
        "algorithmic_bytes_per_launch": bytes_launch, "emit_chobs": bool(emit_chobs),
        "layout_bytes_per_launch": layout_launch,
        "memory": memory_level(N, A, cfg.state_space, B, emit_chobs, 8 if out_dtype == "f64" else 4, packed=bool(code & KERNEL_PACKED)),
        "layout_rate_GBps": layout_launch / (kernel_ms * 1e-3) / 1e9,
        "preroll_slots": preroll, "wall_this_rank": wall_local,
    }
    return env, res


def rollout_sps(device, envs=4096, slots=400, warm=80):
    """SURVEY 8f rows 1 + 3 together, end to end: the driver's slot loop (env step with channel
    observation and state -> reward shaping -> SPS policy picking the next actions), everything
    on the device, three launches per slot.  Not the metric - a reported side measurement."""
    from diral_amd import c2_config
    from diral_amd.driver import DriverLoop
    from diral_amd.sps import SpsPolicy
    cfg = c2_config()
    env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float32, io_ring=2)
    env.reset_topology(seed=GLOBAL_SEED)
    loop = DriverLoop(env, global_reward_avg=True, episode_interval=cfg.episode_interval)
    pol = SpsPolicy(env.B, env.N, env.A, device=device, seed=0)
    loop.bootstrap(pol.prev_action)
    actions = pol.prev_action.clone()
    t0 = None
    for t in range(warm + slots):
        if t == warm:
            torch.cuda.synchronize(device)
            env.metrics(clear=True)
            t0 = time.perf_counter()
        out = loop.slot(actions, t)
        actions = pol.step_from_chobs(env._chobs, actions)
        if out["episode_end"]:
            loop.end_episode()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    env.check()
    m = env.metrics().sum(0)
    return {"what": "examples/rollout_sps.py in short: c2 env (my_step + obtain_state, one launch) + driver reward "
                    "shaping + SPS policy (algorithms/v2x_sps.py) from the channel observation, %d envs, %d slots" % (envs, slots),
            "agent_steps_per_s": envs * env.N * slots / dt, "ms_per_slot": dt / slots * 1e3,
            "collision_fraction": float(m[3] / (m[2] + m[3]))}


def rollout_sps_graph(device, envs=4096, K=48, replays=8):
    """The same closed loop (env step -> reward shaping -> SPS policy) as `rollout_sps`, K slots captured into ONE
    hipGraph and replayed (diral_amd/rollout.py: slot number, policy draws and actions live in device memory; K a
    multiple of 3: the captured step launches keep the slow-env list live, diral_env_set_capture_rotation)."""
    from diral_amd import c2_config
    from diral_amd.rollout import GraphRollout
    from diral_amd.sps import SpsPolicy
    cfg = c2_config()
    env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float32, io_ring=2)
    env.reset_topology(seed=GLOBAL_SEED)
    pol = SpsPolicy(env.B, env.N, env.A, device=device, seed=0)
    ro = GraphRollout(env, pol, K=K)
    ro.run(2)
    torch.cuda.synchronize(device)
    env.metrics(clear=True)
    t0 = time.perf_counter()
    ro.run(replays)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    env.check()
    m = env.metrics().sum(0)
    slots = K * replays
    out = {"what": "c2 env + driver reward shaping + SPS policy, %d envs: %d slots captured into one hipGraph, %d replays" % (envs, K, replays),
           "agent_steps_per_s": envs * env.N * slots / dt, "ms_per_slot": dt / slots * 1e3,
           "collision_fraction": float(m[3] / (m[2] + m[3]))}
    ro.close()
    return out


def rollout_sps_fused(device, envs=4096, slots=400, warm=80, write_chobs=False):
    """The closed loop of `rollout_sps` as ONE launch per slot (`diral_env_step_policy`: env step + reward shaping + SPS
    decision, the channel observation handed over in LDS and - unless `write_chobs` - never written to HBM); eager
    launches from Python, two action buffers alternating.  Same decisions and rewards as the three-launch loop
    (tests/test_gpu_parity.py::test_fused_policy_slot_equals_three_launches)."""
    from diral_amd import c2_config
    from diral_amd.sps import SpsPolicy
    cfg = c2_config()
    env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float32, io_ring=2)
    env.reset_topology(seed=GLOBAL_SEED)
    pol = SpsPolicy(env.B, env.N, env.A, device=device, seed=0)
    acts = [pol.prev_action.clone(), torch.empty_like(pol.prev_action)]
    shaped = [torch.empty((envs, env.N), dtype=torch.float32, device=device) for _ in range(2)]
    sum_r = [torch.empty((envs,), dtype=torch.float32, device=device) for _ in range(2)]
    coll = [torch.empty((envs,), dtype=torch.float32, device=device) for _ in range(2)]
    t0 = None
    for t in range(warm + slots):
        if t == warm:
            torch.cuda.synchronize(device)
            env.metrics(clear=True)
            t0 = time.perf_counter()
        i = t & 1
        env.step_policy(acts[i], t, pol, acts[i ^ 1], shaped_out=shaped[i], sum_r_out=sum_r[i], collision_out=coll[i],
                        global_reward_avg=True, want_chobs=write_chobs)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    env.check()
    from diral_amd.config import KERNEL_POLICY
    assert env.last_kernel() & KERNEL_POLICY
    m = env.metrics().sum(0)
    return {"what": "the closed loop of rollout_sps as ONE launch per slot (diral_env_step_policy: c2 env step + driver reward "
                    "shaping + SPS decision; channel observation %s), %d envs, %d slots" % (
                        "also written to HBM" if write_chobs else "kept on the chip", envs, slots),
            "agent_steps_per_s": envs * env.N * slots / dt, "ms_per_slot": dt / slots * 1e3,
            "collision_fraction": float(m[3] / (m[2] + m[3]))}


def rollout_sps_kslots(device, envs=4096, K=25, launches=16, warm=4, want_obs=False, per_slot_outputs=True):
    """The closed loop of `rollout_sps_fused` with K slots per launch (`diral_env_step_policy`, DiralSlotPolicy::slots = K:
    step_fast64_slots_kernel keeps every env in registers and LDS from slot to slot - no table traffic between slots, no
    histogram unless the last slot's state vector is asked for).  Per-slot shaped rewards / sums / collisions leave as
    [K, ...] arrays; equal to K one-slot launches bit for bit
    (tests/test_gpu_parity.py::test_k_slots_in_one_launch_equal_k_one_slot_launches)."""
    from diral_amd import c2_config
    from diral_amd.sps import SpsPolicy
    cfg = c2_config()
    env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float32, io_ring=2)
    env.reset_topology(seed=GLOBAL_SEED)
    pol = SpsPolicy(env.B, env.N, env.A, device=device, seed=0)
    acts = [pol.prev_action.clone(), torch.empty_like(pol.prev_action)]
    shaped = torch.empty((K, envs, env.N), dtype=torch.float32, device=device) if per_slot_outputs else None
    sum_r = torch.empty((K, envs), dtype=torch.float32, device=device) if per_slot_outputs else None
    coll = torch.empty((K, envs), dtype=torch.float32, device=device) if per_slot_outputs else None
    t0 = None
    for n in range(warm + launches):
        if n == warm:
            torch.cuda.synchronize(device)
            env.metrics(clear=True)
            t0 = time.perf_counter()
        i = n & 1
        env.step_policy(acts[i], n * K, pol, acts[i ^ 1], shaped_out=shaped, sum_r_out=sum_r, collision_out=coll,
                        global_reward_avg=True, slots=K, want_obs=want_obs)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    env.check()
    from diral_amd.config import KERNEL_POLICY
    assert env.last_kernel() & KERNEL_POLICY
    m = env.metrics().sum(0)
    slots = launches * K
    return {"what": "the closed loop of rollout_sps with %d slots per launch (diral_env_step_policy, slots = %d: the env stays on "
                    "the chip from slot to slot; per-slot shaped rewards %s, the last slot's state vector %s), %d envs, %d slots" % (
                        K, K, "written" if per_slot_outputs else "not written", "written" if want_obs else "not computed", envs, slots),
            "agent_steps_per_s": envs * env.N * slots / dt, "ms_per_slot": dt / slots * 1e3,
            "collision_fraction": float(m[3] / (m[2] + m[3]))}


def prefill_kslots(device, K=25, reps=8):
    """The driver's random prefill (main_test.py:99-114: sample -> my_step_design -> obtain_state, every state kept) as ONE
    launch of K slots (`diral_env_prefill`) beside the loop of [diral_env_sample + one fused design step] it replaces, at
    C2's shape; bit-equal to that loop (tests/test_gpu_prefill.py)."""
    from diral_amd import c2_config
    from diral_amd.config import KERNEL_POLICY, STEP_DESIGN
    out = {"what": "random prefill, %d slots per launch (diral_env_prefill) vs the loop of sample + fused my_step_design step; "
                   "us per slot, every slot's state vector written" % K}
    for envs in (1024, 4096):
        env = VecV2VEnv(c2_config(), batch=envs, device=device, out_dtype=torch.float32)
        env.reset_topology(seed=GLOBAL_SEED)
        for t in range(PREROLL):
            env.step(env.sample(t), t)
        nxt = [env.sample(0)]

        def launch(seed):
            _, _, nxt[0] = env.prefill(nxt[0], K, seed)

        def loop(seed):
            for k in range(K):
                env._step(STEP_DESIGN, env.sample(seed + k), 0)

        res = {}
        for name, fn in (("one_launch", launch), ("loop", loop)):
            fn(0)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for r in range(reps):
                fn(1000 * (r + 1))
            torch.cuda.synchronize(device)
            res[name + "_us_per_slot"] = (time.perf_counter() - t0) / (reps * K) * 1e6
            if name == "one_launch":
                assert env.last_kernel() & KERNEL_POLICY
        env.check()
        res["agent_steps_per_s"] = envs * env.N / (res["one_launch_us_per_slot"] * 1e-6)
        out["envs_%d" % envs] = res
        del env
        torch.cuda.empty_cache()
    return out


def c2_graph(device, envs=4096, K=24, replays=42):
    """The headline step (c2: state + reward + channel observation, iid-uniform actions from a ring of K action tensors)
    with K slots captured into ONE hipGraph (slot number on the device: diral_env_set_clock) and replayed: what is left of
    the gap between `ms_per_step` and the kernel time when no launch goes through Python.  Outputs of a replay equal K
    eager steps (tests/test_gpu_parity.py::test_graph_rollout_equals_eager covers the clocked step)."""
    import ctypes
    from diral_amd import c2_config
    from diral_amd.config import STEP_MY_STEP
    from diral_amd.rollout import SlotClock
    cfg = c2_config()
    env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float32, io_ring=2)
    env.reset_topology(seed=GLOBAL_SEED)
    clock = SlotClock(device)
    env.set_clock(clock.t)                      # (the env keeps the tensor alive: VecV2VEnv.set_clock)
    acts = [env.sample(seed=100 + i) for i in range(K)]

    def k_slots():
        for k in range(K):
            env._step(STEP_MY_STEP, acts[k], k, want_chobs=True)
        st = env.lib.diral_clock_add(ctypes.c_void_p(clock.ptr()), K, env._stream())
        assert st == 0, st

    for _ in range(4):
        k_slots()                               # past the ghost-entry phase; lazy conversions done before the capture
    torch.cuda.synchronize(device)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    assert K % 3 == 0                           # the captured launches rotate the slow-env sets: VecV2VEnv.set_capture_rotation
    env.set_capture_rotation(True)
    from diral_amd.rollout import no_finalizers_during_capture
    with no_finalizers_during_capture(), torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            k_slots()
    env.set_capture_rotation(False)
    torch.cuda.current_stream(device).wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(replays):
        graph.replay()
    ev1.record()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    env.check()
    env.set_clock(None)
    slots = K * replays
    return {"what": "the bench step (c2, %d envs), %d slots per hipGraph (device slot clock, ring of %d action tensors), %d replays"
                    % (envs, K, K, replays),
            "ms_per_step": dt / slots * 1e3, "ms_per_step_events": ev0.elapsed_time(ev1) / slots,
            "agent_steps_per_s": envs * env.N * slots / dt}


def secondary_modes(device, envs=4096, slots=200, warm=100):
    """the secondary observation modes of obtain_state at the c2 shapes - the
    step on the specialised kernels plus the observation launch of csrc/posdist_kernel.hpp - per slot,
    HIP events.  A reported side measurement (profiles/secondary_modes.py is the long form)."""
    from diral_amd.config import bench_config
    out = {}
    for key, st in (("sorted_distances", dict(add_positional_dist=True)), ("type1_histogram", dict(add_positional_dist_type=1))):
        cfg = bench_config(64, 32, 2000.0, State=st)
        env = VecV2VEnv(cfg, batch=envs, device=device, out_dtype=torch.float32)
        env.reset_topology(seed=GLOBAL_SEED)
        acts = [env.sample(seed=i) for i in range(16)]
        for t in range(warm):
            env.step(acts[t % 16], t)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for t in range(warm, warm + slots):
            env.step(acts[t % 16], t)
        ev1.record()
        torch.cuda.synchronize(device)
        env.check()
        ms = ev0.elapsed_time(ev1) / slots
        out[key] = {"workload": "c2 shapes, batch=%d, State %s" % (envs, st), "state_space": cfg.state_space,
                    "ms_per_slot": ms, "agent_steps_per_s": envs * 64 / (ms * 1e-3)}
        del env
    return out


def streamed_c2(device, groups, steps=600, warm=100):
    """The c2 batch stepped as `groups` sub-batches on as many HIP streams (diral_amd/streamed.py): slot t + 1 of a
    sub-batch only waits for slot t of the SAME sub-batch, so the tail of one launch overlaps with the head of the next.
    Every env takes `steps` slots; state + reward + channel observation; wall time per slot of ALL envs.  A side
    measurement: the headline stays the single launch whose outputs are complete, in stream order, after every step."""
    from diral_amd.streamed import StreamedVecEnv
    N, A, L, B, _ = WORKLOADS["c2"]
    cfg = bench_config(N, A, L)
    env = StreamedVecEnv(cfg, batch=B, groups=groups, device=device, out_dtype=torch.float32)
    env.reset_topology(seed=GLOBAL_SEED)
    acts = [env.sample(seed=1000 + i) for i in range(32)]
    torch.cuda.synchronize(device)          # the pre-generated actions are complete: no per-slot hand-shake needed
    for t in range(warm):
        env.step(acts[t % 32], t, sync=False, actions_ready=True)
    env.wait()
    torch.cuda.synchronize(device)
    # three timed runs, the median reported: with G launches per slot the host (~10 us per launch from Python) runs close
    # to the device's pace, and one hiccup of the host shows in a single run
    runs, t = [], warm
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps // 3):
            env.step(acts[t % 32], t, sync=False, actions_ready=True)
            t += 1
        env.wait()
        torch.cuda.synchronize(device)
        runs.append((time.perf_counter() - t0) / (steps // 3))
    env.check()
    dt = sorted(runs)[1]
    out = {"workload": "c2: %d-UE/%d-res, batch=%d as %d sub-batches on %d streams (no per-slot join)" % (N, A, B, groups, groups),
           "agent_steps_per_s": B * N / dt, "ms_per_step": dt * 1e3, "ms_per_step_runs": [r * 1e3 for r in runs]}
    env.close()
    return out


def streamed_c2_sets(device, groups, sets=4):
    """`streamed_c2` on `sets` different sets of pooled streams, the fastest set reported and every set listed.  Which hardware
    queue a stream lands on is the runtime's bookkeeping and depends on the process's earlier stream users:
    two streams of a set may share a queue, and their launches then run back to back (G = 2: 85 instead of 50 us per slot).
    The sub-batch design is measured by the set whose streams got queues of their own."""
    runs = [streamed_c2(device, groups, steps=300, warm=60) for _ in range(sets)]
    best = min(runs, key=lambda r: r["ms_per_step"])
    best = dict(best)
    best["ms_per_step_by_stream_set"] = [r["ms_per_step"] for r in runs]
    med = sorted(r["ms_per_step"] for r in runs)
    best["ms_per_step_median_of_sets"] = 0.5 * (med[(len(med) - 1) // 2] + med[len(med) // 2])
    best["agent_steps_per_s_median_of_sets"] = best["agent_steps_per_s"] * best["ms_per_step"] / best["ms_per_step_median_of_sets"]
    best["note"] = ("fastest of %d stream sets, the median of the sets beside it (a set whose streams share a hardware queue "
                    "serialises: runtime bookkeeping, DESIGN.md 3.6)" % sets)
    return best


def short(res):
    """The keys of a secondary measurement that go into the JSON line (frac: this layout's compulsory
    bytes over the kernel time over the HBM peak, as in the main roofline object)."""
    return {"workload": "%s: %d-UE/%d-res, batch=%d" % (res["workload"], res["N"], res["A"], res["B"]),
            "agent_steps_per_s": res["agent_steps_per_s"], "ms_per_step": res["ms_per_step"],
            "kernel_ms": res["kernel_ms"], "kernel": res["kernel"],
            "bytes_per_launch": res["layout_bytes_per_launch"],
            "achieved_GBps": res["layout_rate_GBps"], "frac": res["layout_rate_GBps"] / HBM_PEAK_GBPS,
            "model_throughput_TBps": res["algorithmic_bytes_per_launch"] / (res["kernel_ms"] * 1e-3) / 1e12}


def self_launch(gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: N ranks of this script under torch.distributed.run."""
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    msg = check_visible_gpus(gpus, visible)
    if msg:
        print("bench.py: %s" % msg, file=sys.stderr)
        return 2
    return spawn_ranks(os.path.abspath(__file__), sys.argv[1:], gpus)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="envs per GPU (default: the workload's)")
    ap.add_argument("--out-dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--sticky", type=float, default=0.0,
                    help="probability an agent keeps its resource (0 = iid uniform, worst case)")
    ap.add_argument("--step-mode", default="my_step", choices=["my_step", "my_step_ch"],
                    help="my_step = the metric's step kind; my_step_ch = the PRR-reward variant (test_env.py:351-443)")
    ap.add_argument("--emit-chobs", type=int, default=1, choices=[0, 1],
                    help="1 (default): the timed step also writes the channel observation `obs` of the reference "
                         "step (4*N*A bytes per env-slot, part of SURVEY 8d's byte model); 0: state + reward only")
    ap.add_argument("--no-cpu-baseline", "--lean", dest="lean", action="store_true",
                    help="only the timed run (profiling passes): no CPU baseline, PRR parity or secondary workloads")
    args = ap.parse_args()

    if not under_launcher():
        if args.gpus > 1:
            return self_launch(args.gpus)                # N ranks of this script, one per GPU
        if args.gpus < 1:
            print("bench.py: --gpus must be >= 1", file=sys.stderr)
            return 2
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world), file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    msg = check_visible_gpus(local_rank + 1, torch.cuda.device_count())
    if msg:
        print("bench.py: rank %d (local rank %d): %s" % (rank, local_rank, msg), file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the RCCL group is always created - also for
    # one rank, so the single-GPU run exercises the same collective code path
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        if rank == 0:
            print("bench.py: RCCL process group up: backend=%s world=%d (rank 0 on %s)" % (
                dist.get_backend(), dist.get_world_size(), torch.cuda.get_device_name(device)), file=sys.stderr)

    emit = bool(args.emit_chobs)
    env, res = run_workload(args.workload, device, rank, world, args.steps, args.warmup, args.batch, args.out_dtype,
                            args.step_mode, emit, args.sticky, use_dist)
    totals = gather_metrics(env)                         # RCCL all-reduce (metrics only)
    per_rank = None
    if use_dist:
        # every rank's own wall time and kernel time of the timed region (value uses the MAX wall)
        mine = torch.tensor([res["wall_this_rank"] / args.steps * 1e3, res["kernel_ms"]], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ms = [float(t[0].item()) for t in allr]
        km = [float(t[1].item()) for t in allr]
        per_rank = {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "kernel_ms_min": min(km), "kernel_ms_max": max(km),
                    "ms_per_step": ms}
    cfg = res["cfg"]
    del env
    if use_dist:
        # the collective part of the job is over: the other ranks leave, rank 0 stays for the host-side legs
        # (CPU baseline, PRR parity) so that nobody sits in a collective while the host cores are being timed
        dist.barrier()
        torch.cuda.synchronize(device)
        dist.destroy_process_group()
    if rank != 0:
        return 0

    N, A, B, L = res["N"], res["A"], res["B"], res["L"]
    tkey = args.workload + ("" if emit else "_nochobs")
    pmc = load_pmc(tkey if (args.batch in (0, WORKLOADS[args.workload][3]) and args.out_dtype == "f32"
                            and args.step_mode == "my_step" and args.sticky == 0) else "")
    line = {
        "metric": "agent-steps/sec (envs x vehicles), %d-UE/%d-res batched env" % (N, A),
        "value": res["agent_steps_per_s"],
        "unit": "agent-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s: %d-UE/%d-res, batch=%d envs per GPU x %d GPU, L=%g m, Rc=%g, K=%d bins, "
                        "reward_design=2, %s+obtain_state fused (state + reward%s), %s actions, %s outputs" % (
                            args.workload, N, A, B, world, L, cfg.communication_range,
                            cfg.State.num_bins, args.step_mode, " + channel observation" if emit else "",
                            "iid-uniform" if args.sticky == 0 else "sticky(p=%.2f)" % args.sticky, args.out_dtype),
            "batch_per_gpu": B, "num_users": N, "num_channels": A, "state_space": cfg.state_space,
            "emit_chobs": emit, "preroll_slots": res["preroll_slots"],
            "parallelism": "env-shard x%d (no data-path collective; one global seed, rank r = envs [r*B, (r+1)*B))" % world,
            "launcher": "torch.distributed.run, RCCL group of %d" % world if use_dist else "single process, no process group",
        },
        "roofline": roofline_object(res, pmc),
        "episode_metrics": totals,
    }
    if per_rank:
        line["per_rank"] = per_rank
    if world == 1 and not args.lean:
        # short same-run measurements of the other BASELINE.json configurations and of the
        # opposite --emit-chobs setting (each: PREROLL untimed slots + a few timed ones)
        also = {}
        specs = [("c2_emit_chobs_%d" % (0 if emit else 1), "c2", dict(emit_chobs=not emit, steps=300)),
                 ("c4shard", "c4shard", dict(emit_chobs=emit, steps=100)),
                 ("c3", "c3", dict(emit_chobs=emit, steps=40)),
                 ("c5", "c5", dict(emit_chobs=emit, steps=60)),
                 ("n1024_three_launch_form", "n1024", dict(emit_chobs=emit, steps=10))]
        for key, wl, kw in specs:
            if args.workload != "c2" or args.batch:
                break
            try:
                e2, r2 = run_workload(wl, device, 0, 1, kw["steps"], 10, 0, args.out_dtype, args.step_mode,
                                      kw["emit_chobs"], args.sticky, False)
                del e2
                torch.cuda.empty_cache()
                also[key] = short(r2)
                also[key]["emit_chobs"] = kw["emit_chobs"]
            except Exception as exc:                      # a secondary measurement never fails the line
                also[key] = {"error": repr(exc)[:200]}
        if args.workload == "c2" and not args.batch:
            try:
                # SURVEY 8d's "converged" action distribution: each agent keeps its resource with p = 0.9
                e2, r2 = run_workload("c2", device, 0, 1, 300, 10, 0, args.out_dtype, args.step_mode, emit, 0.9, False)
                del e2
                also["c2_sticky_0.9"] = short(r2)
                also["c2_sticky_0.9"]["emit_chobs"] = emit
                if args.out_dtype == "f32":
                    # float64 outputs: the reference's own array dtype (state, reward, channel observation bit for bit)
                    e2, r2 = run_workload("c2", device, 0, 1, 300, 10, 0, "f64", args.step_mode, emit, args.sticky, False)
                    del e2
                    also["c2_f64"] = short(r2)
                    also["c2_f64"]["emit_chobs"] = emit
                    also["c2_f64"]["out_dtype"] = "f64"
                also["c2_streams2"] = streamed_c2_sets(device, 2)
                also["c2_streams4"] = streamed_c2_sets(device, 4)
                torch.cuda.empty_cache()
                also["rollout_sps"] = rollout_sps(device)
                torch.cuda.empty_cache()
                also["rollout_sps_graph"] = rollout_sps_graph(device)
                also["rollout_sps_fused"] = rollout_sps_fused(device)
                also["rollout_sps_fused_chobs"] = rollout_sps_fused(device, write_chobs=True)
                also["rollout_sps_kslots"] = rollout_sps_kslots(device)
                also["rollout_sps_kslots_state"] = rollout_sps_kslots(device, K=5, launches=80, warm=16, want_obs=True)
                also["prefill_kslots"] = prefill_kslots(device)
                also["c2_graph"] = c2_graph(device)
                torch.cuda.empty_cache()
                also["secondary_observation_modes"] = secondary_modes(device)
                torch.cuda.empty_cache()
            except Exception as exc:
                also["rollout_sps"] = {"error": repr(exc)[:200]}
        line["also_measured"] = also
    if not args.lean:
        # host-side legs, on rank 0 at every N (the other ranks have left the job by now)
        line["cpu_baseline"] = cpu_baseline(cfg)
        line["prr_parity"] = prr_parity(cfg, device)
    print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
