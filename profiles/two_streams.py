#!/usr/bin/env python3
"""The c2 batch (4096 envs) as G independent sub-batches on G HIP streams: launch t + 1 of a sub-batch only depends
on launch t of the SAME sub-batch, so the tail of one launch (4096 workgroups on 256 CUs x 7 resident = 2.29 rounds)
overlaps with the head of the next.  Wall time per slot of ALL envs, state + reward + channel observation."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diral_amd.config import c2_config, STEP_MY_STEP
from diral_amd.vec_env import VecV2VEnv

B = int(os.environ.get("B", 4096))
for G in (1, 2, 4):
    cfg = c2_config()
    envs, acts, streams = [], [], []
    for g in range(G):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            e = VecV2VEnv(cfg, batch=B // G, device="cuda:0", out_dtype=torch.float32, env_offset=g * (B // G))
            e.reset_topology(seed=1234)
            acts.append([e.sample(seed=1000 + i) for i in range(16)])
        envs.append(e); streams.append(s)
    torch.cuda.synchronize()

    def run(n, t0):
        for t in range(t0, t0 + n):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    envs[g]._step(STEP_MY_STEP, acts[g][t % 16], t, want_chobs=True)
    run(100, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(600, 100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("G=%d sub-batches of %d envs: %.2f us per slot of all %d envs (%.3g agent-steps/s)" % (G, B // G, dt / 600 * 1e6, B, B * 64 * 600 / dt))
