#!/usr/bin/env python3
"""How many 8-column passes of a C5-shaped handle (128 UE / 64 res, dynamic density) hold an entry 7 or more stamps behind its
subject - the passes the packed table form sends through the planes (csrc/step_wide.hpp, flagged passes) - as the rollout ages."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diral_amd.config import bench_config
from diral_amd.vec_env import VecV2VEnv
N, A, L = 128, 64, 4000.0
cfg = bench_config(N, A, L, mobility_vary=True)
B = 64
env = VecV2VEnv(cfg, batch=B, out_dtype=torch.float32)
env.reset_topology(seed=1234)
ei = cfg.episode_interval
marks = (64, 150, 300, 600, 1200, 2400)
for t in range(marks[-1] + 1):
    env.step(env.sample(seed=1000 + t % 32), t)
    if t % ei == ei - 1:
        env.update_velocity(seed=t)
    if t in marks:
        st = env.export_state()
        seq = st["seq"].cpu().numpy().astype(np.int64)          # [B][subject][viewer]
        own = np.stack([np.diagonal(seq[b]) for b in range(B)])
        lag = own[:, :, None] - seq
        heard = seq != 0
        far = heard & (lag >= 7)
        per_subj = far.any(axis=2)                               # [B][subject]
        p8 = per_subj.reshape(B, N // 8, 8).any(axis=2)
        wg = p8.any(axis=1)
        # how many DIFFERENT sequence numbers the far entries of a subject hold (1: a cluster that converged on the last stamp
        # that crossed - the coded pass could carry those as one extra level)
        seqf = np.where(far, seq, 0)
        srt = np.sort(seqf, axis=2)
        distinct = ((srt[:, :, 1:] != srt[:, :, :-1]) & (srt[:, :, 1:] != 0)).sum(axis=2) + (srt[:, :, 0] != 0)
        d8 = distinct.reshape(B, N // 8, 8).max(axis=2)
        fl = p8
        print("        flagged passes whose subjects hold <= 1 far number: %.3f, <= 2: %.3f; distinct per flagged subject: mean %.2f max %d"
              % ((d8[fl] <= 1).mean() if fl.any() else 1.0, (d8[fl] <= 2).mean() if fl.any() else 1.0,
                 distinct[per_subj].mean() if per_subj.any() else 0.0, int(distinct.max())))
        print("t=%5d  entries >= 7 behind: %.4f  never heard: %.4f  subjects with one: %.3f  8-column passes: %.3f  envs with a flagged pass: %.3f  passes per env: %.2f"
              % (t, far.mean(), (~heard).mean(), per_subj.mean(), p8.mean(), wg.mean(), p8.sum(1).mean()))
