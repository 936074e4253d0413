#!/usr/bin/env python3
"""Step time of the secondary observation modes (add_positional_dist,
add_positional_dist_type 1) next to the metric's type-2 piggybacked histogram, C2 shapes.

  python profiles/secondary_modes.py            # on an MI355X
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diral_amd.config import bench_config  # noqa: E402
from diral_amd.vec_env import VecV2VEnv  # noqa: E402

B = int(os.environ.get("B", 4096))
SHAPES = {"c2": (64, 32, 2000.0), "c5": (128, 64, 4000.0), "c3": (256, 64, 4000.0)}   # batch B, B / 2, B / 4
MODES = {
    "type-2 piggy histogram (metric)": {},
    "add_positional_dist (sorted true distances)": dict(add_positional_dist=True, add_positional_dist_piggy=False),
    "type-1 piggy histogram": dict(add_positional_dist_type=1),
    "sorted distances + type-2 histogram": dict(add_positional_dist=True),
}
ONLY = os.environ.get("MODE_FILTER", "")              # substring of the mode names to run (profiling passes)
SLOTS = int(os.environ.get("SLOTS", 300))
for wl in os.environ.get("WORKLOADS", "c2").split(","):
    N, A, L = SHAPES[wl]
    for name, st in MODES.items():
        if ONLY and ONLY not in name:
            continue
        cfg = bench_config(N, A, L, State=st) if st else bench_config(N, A, L)
        env = VecV2VEnv(cfg, batch={"c2": B, "c5": B // 2, "c3": B // 4}[wl], out_dtype=torch.float32)
        env.reset_topology(seed=1)
        acts = [env.sample(seed=i) for i in range(32)]
        for t in range(200):
            env.step(acts[t % 32], t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(200, 200 + SLOTS):
            env.step(acts[t % 32], t)
        e1.record()
        torch.cuda.synchronize()
        print("%s %-46s S=%4d  %.1f us / slot  kernel code %d" % (wl, name, cfg.state_space, e0.elapsed_time(e1) / SLOTS * 1e3,
                                                                   env.last_kernel()))
        del env
