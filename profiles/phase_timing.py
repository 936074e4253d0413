#!/usr/bin/env python3
"""Per-phase shader-clock timing of the fused step kernel (DIRAL_TIMING build).

  DIRAL_LIB=diral_amd/variants/timing.so python profiles/phase_timing.py

Stamps (s_memtime, per wave): 0 start, 1 end P0, 2 end P1, 3 after barrier + P2,
4 end merge, 5 end finalize/hist, 6 after barrier, 7 end of P4 (kernel end).
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diral_amd.config import bench_config, c2_config  # noqa: E402
from diral_amd.vec_env import VecV2VEnv  # noqa: E402

B = int(os.environ.get("B", 4096))
WL = os.environ.get("WORKLOAD", "c2")
cfg = {"c2": c2_config(), "c3": bench_config(256, 64, 4000.0), "c5": bench_config(128, 64, 4000.0)}[WL]
NW = {"c2": 4, "c3": 8 if not os.environ.get("DIRAL_NO_WIDE") else 16, "c5": 8}[WL]
GENERAL = WL != "c2"
env = VecV2VEnv(cfg, batch=B, out_dtype=torch.float32)
env.reset_topology(seed=1)
STICKY = float(os.environ.get("STICKY", "0"))            # probability an agent keeps its resource (converged policy)
acts = [env.sample(seed=i) for i in range(64)]
if STICKY > 0:
    for i in range(1, 64):
        keep = torch.rand(acts[i].shape, device=acts[i].device) < STICKY
        acts[i] = torch.where(keep, acts[i - 1], acts[i])
for t in range(64):
    env.step(acts[t], t)
torch.cuda.synchronize()
fn = env.lib.diral_env_debug_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = np.zeros((B, NW, 8), np.uint64)
assert fn(env._h, buf.ctypes.data_as(ctypes.c_void_p), NW) == 0
t = buf.astype(np.int64)
names = ["P0 loads/init", "P1 closest-tx", "barrier+P2", "P3 merge", "P3 finalize+hist", "wait barrier", "P4 output"]
WIDE = GENERAL and not os.environ.get("DIRAL_NO_WIDE")
if WIDE:      # step_wide.hpp: load / merge / finalize times are accumulated over the passes
    names = ["P0 load+barrier", "P1 closest-tx", "barrier+P2", "P3 table loads+stamp+ranks", "P3 merge",
             "P3 finalize+hist", "barrier+cnt+P4"]
elif GENERAL:   # step_kernel.hpp stamps (merge/finalize of the LAST pass only)
    names = ["P0 load+barrier", "P1 closest-tx", "wait barrier", "P2 rewards", "P3 all passes but last finalize",
             "P3 last finalize+hist", "wait barrier"]
d = np.diff(t, axis=2)
print("B=%d; mean cycles per wave per phase (s_memtime ticks), by wave:" % B)
for i, n in enumerate(names):
    print("  %-32s %s   all=%.0f" % (n, " ".join("%7.0f" % d[:, w, i].mean() for w in range(min(NW, 4))), d[:, :, i].mean()))
life = (t[:, :, 7] - t[:, :, 0])
if os.environ.get("SLOW_SPLIT"):
    # workgroups whose waves live more than twice the median (step_wide, packed form: envs whose passes left the codes)
    wl = life.mean(axis=1)
    slow = wl > 2 * np.median(wl)
    for nm, sel in (("slow", slow), ("others", ~slow)):
        if sel.any():
            print("  %s workgroups (%d): %s  lifetime %.0f" % (nm, int(sel.sum()), " | ".join("%s %.0f" % (n.split()[0] + n.split()[1][:6] if len(n.split()) > 1 else n, d[sel][:, :, i].mean()) for i, n in enumerate(names)), wl[sel].mean()))
print("  wave lifetime: mean %.0f  p50 %.0f  p99 %.0f" % (life.mean(), np.median(life), np.percentile(life, 99)))
span = t[:, :, 7].max() - t[:, :, 0].min()
print("  kernel span %d ticks" % span)
start = t[:, 0, 0] - t[:, 0, 0].min()
print("  WG start time percentiles (ticks):", [int(np.percentile(start, q)) for q in (0, 25, 50, 75, 100)])
# occupancy timeline per XCD (the s_memtime counters of different XCDs are not aligned): HW_ID / XCC_ID of every wave
# sit behind the stamps (c2 / DIRAL_TIMING build only)
if WL == "c2":
    full = np.zeros((B * 4096,), np.uint64)
    assert fn(env._h, full.ctypes.data_as(ctypes.c_void_p), 512) == 0
    ids = full[B * 32:B * 32 + B * 4].reshape(B, 4)
    hw = (ids & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (ids >> np.uint64(32)).astype(np.int64) & 15
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    sh = (hw >> 12) & 1
    print("  XCC ids %s  SE ids %s  SH ids %s  CU ids %s" % (sorted(set(xcc[:, 0].tolist())), sorted(set(se[:, 0].tolist())),
                                                             sorted(set(sh[:, 0].tolist())), sorted(set(cu[:, 0].tolist()))))
    ws = t[:, :, 0].min(axis=1)
    we = t[:, :, 7].max(axis=1)
    grp = xcc[:, 0] * 8 + se[:, 0]                 # s_memtime is per (XCC, shader engine)
    for x in sorted(set(grp.tolist()))[:3]:
        idx = np.nonzero(grp == x)[0]
        t0 = ws[idx].min()
        s_ = (ws[idx] - t0).astype(np.float64)
        e_ = (we[idx] - t0).astype(np.float64)
        sp = e_.max()
        bins = np.linspace(0, sp, 25)
        alive = [int(((s_ < hi) & (e_ > lo)).sum()) for lo, hi in zip(bins[:-1], bins[1:])]
        ncu = len(set(zip(se[idx, 0].tolist(), sh[idx, 0].tolist(), cu[idx, 0].tolist())))
        print("  XCC*8+SE %d: %d WGs on %d CUs (WG ids mod 8: %s); span %d ticks; alive per 1/24 of the span: %s" % (
            x, len(idx), ncu, sorted(set((idx % 8).tolist())), sp, alive))
        print("     WG start percentiles %s\n     end percentiles %s" % (
            [int(np.percentile(s_, q)) for q in (0, 10, 25, 50, 75, 90, 100)], [int(np.percentile(e_, q)) for q in (0, 10, 25, 50, 75, 90, 100)]))
        per_cu = {}
        for i in idx:
            per_cu.setdefault((int(se[i, 0]), int(sh[i, 0]), int(cu[i, 0])), []).append(i)
        print("     WGs per CU: min %d max %d" % (min(len(v) for v in per_cu.values()), max(len(v) for v in per_cu.values())))
