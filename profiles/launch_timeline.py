#!/usr/bin/env python3
"""When do the workgroups of a launch start and end, on the chip-wide clock (DIRAL_TIMING build)?

  DIRAL_LIB=variants_tmp/lib_timing.so B=4096 python profiles/launch_timeline.py

Every workgroup of the C2 step kernel records s_memrealtime (100 MHz, one clock for the whole chip) at its start and
at its end, two consecutive launches into two sets: start skew of a launch, how long it takes to drain, and the gap
between the last workgroup of launch t and the first of launch t + 1 on the same stream.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diral_amd.config import c2_config  # noqa: E402
from diral_amd.vec_env import VecV2VEnv  # noqa: E402

B = int(os.environ.get("B", 4096))
env = VecV2VEnv(c2_config(), batch=B, out_dtype=torch.float32)
env.reset_topology(seed=1)
acts = [env.sample(seed=i) for i in range(64)]
STICKY = float(os.environ.get("STICKY", "0"))            # probability an agent keeps its resource (converged policy)
if STICKY > 0:
    for i in range(1, 64):
        keep = torch.rand(acts[i].shape, device=acts[i].device) < STICKY
        acts[i] = torch.where(keep, acts[i - 1], acts[i])
for t in range(64):
    env._step(0, acts[t], t, want_chobs=True)
torch.cuda.synchronize()
fn = env.lib.diral_env_debug_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
full = np.zeros((B * 4096,), np.uint64)
assert fn(env._h, full.ctypes.data_as(ctypes.c_void_p), 512) == 0
rt = full[B * 40:B * 44].astype(np.float64).reshape(2, B, 2) * 10e-3     # us; set = slot parity: [0] = slot 62, [1] = slot 63
prev, last = rt[0], rt[1]
ok = (last[:, 0] > 0) & (last[:, 1] > 0) & (prev[:, 0] > 0) & (prev[:, 1] > 0)
if not ok.all():
    bad = np.nonzero(~ok)[0]
    print("  (%d workgroups without a complete pair of stamps are left out: envs %s ...; zero fields last start %d end %d prev start %d end %d)" % (
        len(bad), bad[:8].tolist(), int((last[:, 0] <= 0).sum()), int((last[:, 1] <= 0).sum()), int((prev[:, 0] <= 0).sum()), int((prev[:, 1] <= 0).sum())))
    keep = np.nonzero(ok)[0]
    last, prev = last[keep], prev[keep]
    B = len(keep)
t0 = last[:, 0].min()
st = np.sort(last[:, 0] - t0)
en = np.sort(last[:, 1] - t0)
print("B=%d: launch span (first start -> last end) %.2f us; previous launch %.2f us" % (
    B, en[-1], prev[:, 1].max() - prev[:, 0].min()))
print("  gap: last end of the previous launch -> first start of this one %.2f us" % (t0 - prev[:, 1].max()))
q = (0, 10, 25, 50, 75, 90, 99, 100)
print("  start of workgroups after the first, us, percentiles %s: %s" % (q, " ".join("%.2f" % np.percentile(st, x) for x in q)))
print("  end of workgroups, us:                                    %s" % " ".join("%.2f" % np.percentile(en, x) for x in q))
life = last[:, 1] - last[:, 0]
print("  workgroup lifetime us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (
    life.mean(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max()))
# how many workgroups are alive over time
edges = np.linspace(0, en[-1], 33)
alive = [int(((last[:, 0] - t0 < hi) & (last[:, 1] - t0 > lo)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
print("  alive per 1/32 of the span: %s" % alive)
# by dispatch order: start time of workgroup b
idx = np.arange(B)
for lo in range(0, B, max(1, B // 8)):
    sl = slice(lo, lo + max(1, B // 8))
    print("  workgroups %5d..%5d: start %.2f..%.2f  end %.2f..%.2f" % (lo, min(B, lo + max(1, B // 8)) - 1, (last[sl, 0] - t0).min(),
                                                                 (last[sl, 0] - t0).max(), (last[sl, 1] - t0).min(), (last[sl, 1] - t0).max()))
# which path the slow workgroups took (bits 0-3: keyed quads, 4-7: quads on the general column loop), per wave
path = full[env.B * 48:env.B * 52].astype(np.int64).reshape(env.B, 4)
if not ok.all():
    path = path[keep]
keyed = np.array([[bin(int(x) & 15).count("1") for x in row] for row in path]).sum(axis=1)
gen = np.array([[bin((int(x) >> 4) & 15).count("1") for x in row] for row in path]).sum(axis=1)
order = np.argsort(-life)
print("  workgroups with keyed quads: %d (%.1f %%), with general-loop quads: %d (%.1f %%)" % (
    (keyed > 0).sum(), 100.0 * (keyed > 0).mean(), (gen > 0).sum(), 100.0 * (gen > 0).mean()))
print("  slowest 12 workgroups: " + "; ".join("%.1f us k%d g%d" % (life[i], keyed[i], gen[i]) for i in order[:12]))
for lo, hi in ((0, 0), (1, 4), (5, 12), (13, 16)):
    sel = (keyed >= lo) & (keyed <= hi) & ((gen == 0) | (lo > 0))
    if sel.any():
        print("  keyed quads %d..%d: %d workgroups, lifetime mean %.2f us  max %.2f" % (lo, hi, sel.sum(), life[sel].mean(), life[sel].max()))
sel = (keyed == 0) & (gen > 0)
if sel.any():
    print("  no keyed quad, general-loop quads: %d workgroups, lifetime mean %.2f us  max %.2f" % (sel.sum(), life[sel].mean(), life[sel].max()))
# the slow-first sets as the last launch left them (timing builds export them)
try:
    fs = env.lib.diral_env_debug_slow
    fs.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    sbuf = np.zeros((2 * (env.B + env.B // 4 + 64 + 32),), np.uint32)
    w = fs(env._h, sbuf.ctypes.data_as(ctypes.c_void_p))
    if w > 0:
        smax = max(16, min(4096, env.B >> 2))
        for i in range(2):
            st = sbuf[i * w:(i + 1) * w]
            tag = int(st[0]) | (int(st[1]) << 32)
            cnt = int(st[2]) | (int(st[3]) << 32)
            place = st[16 + smax:16 + smax + env.B]
            print("  set %d: clock %d writer id %d; count clock %d entries %d; places != 0: %d (first %s)" % (
                i, tag >> 32, tag & 0xffffffff, cnt >> 32, cnt & 0xffffffff, int((place != 0).sum()), place[:6].tolist()))
except AttributeError:
    pass
