#!/usr/bin/env python3
"""Per-phase shader-clock timing of the LAST slot of a K-slot launch (step_fast64_slots_kernel, DIRAL_TIMING build) beside the
one-slot fused kernel:   DIRAL_LIB=variants_tmp/lib_kst.so python profiles/kslots_timing.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diral_amd import c2_config  # noqa: E402
from diral_amd.sps import SpsPolicy  # noqa: E402
from diral_amd.vec_env import VecV2VEnv  # noqa: E402

B = int(os.environ.get("B", 4096))
names = ["P0", "P1", "barrier+P2", "stamp+merge", "P3b", "barrier", "P4/shape"]
for K in (1, 25):
    env = VecV2VEnv(c2_config(), batch=B, out_dtype=torch.float32)
    env.reset_topology(seed=1)
    pol = SpsPolicy(env.B, env.N, env.A, seed=0)
    a, n = pol.prev_action.clone(), torch.empty_like(pol.prev_action)
    t = 0
    for _ in range(60 if K == 1 else 3):
        env.step_policy(a, t, pol, n, slots=K, want_obs=(K == 1))
        a, n = n, a
        t += K
    torch.cuda.synchronize()
    fn = env.lib.diral_env_debug_timing
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros((B, 4, 8), np.uint64)
    assert fn(env._h, buf.ctypes.data_as(ctypes.c_void_p), 4) == 0
    d = np.diff(buf.astype(np.int64), axis=2)
    print("K=%d, B=%d: mean cycles per wave per phase of the last slot" % (K, B))
    for i, nm in enumerate(names):
        if K > 1 and i == 0:
            continue                                  # (stamp 0 is the kernel's start: not a phase of the last slot)
        print("  %-14s %s" % (nm, " ".join("%7.0f" % d[:, w, i].mean() for w in range(4))))
    print("  slot (stamps 1 -> 7): %.0f" % (buf[:, :, 7].astype(np.int64) - buf[:, :, 1].astype(np.int64)).mean())
    life = (buf[:, :, 7].astype(np.int64) - buf[:, :, 0].astype(np.int64))
    print("  workgroup life / K: mean %.0f  p50 %.0f  p99 %.0f cycles per slot" % (life.mean() / K, np.median(life) / K, np.percentile(life, 99) / K))
