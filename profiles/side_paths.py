#!/usr/bin/env python3
"""Per-call time (HIP events) of the paths beside the headline step, at the bench shapes:
stand-alone obtain_state (diral_env_observe -> observe_kernel.hpp), the table-less step (a State block without
piggybacked tables, test_env.py:138-139, 231-238), my_step with PRR metrics (DIRAL_F_TRACK_PRR) and a static
topology - each on the dispatch's choice and on the general kernel.   python profiles/side_paths.py [c2|c3|c5]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diral_amd.config import bench_config
from diral_amd.vec_env import VecV2VEnv

SHAPES = {"c2": (64, 32, 2000.0, 4096), "c3": (256, 64, 4000.0, 1024), "c5": (128, 64, 4000.0, 2048)}


def timed(fn, n=100, warm=30):
    for t in range(warm):
        fn(t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(warm, warm + n):
        fn(t)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for name in (sys.argv[1:] or ["c2", "c3", "c5"]):
        N, A, L, B = SHAPES[name]
        print("== %s shapes: N=%d A=%d B=%d" % (name, N, A, B))
        for label, state, extra in (("plain", {}, {}), ("no tables", dict(add_positional_dist_piggy=False), {}),
                                    ("track_prr", {}, dict(track_prr=True)),
                                    ("static", {}, dict(mobility=False, enable_design_topology=True))):
            cfg = bench_config(N, A, L, State=state, **extra)
            for general in (False, True):
                env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32)
                env.reset_topology(seed=1)
                env.force_general_kernel(general)
                acts = [env.sample(seed=i) for i in range(8)]
                us = timed(lambda t: env.step(acts[t % 8], t))
                k = env.last_kernel()
                fc = torch.rand((B, N, A), dtype=torch.float64, device="cuda:0")
                fr = torch.rand((B, N), dtype=torch.float64, device="cuda:0")
                uo = timed(lambda t: env.obtain_state(fc, acts[t % 8], fr))
                print("  %-10s %-8s step %8.1f us (kernel code %3d)   stand-alone obtain_state %8.1f us (code %d)" % (
                    label, "general" if general else "dispatch", us, k, uo, env.last_kernel()))
                env.check()
                del env


if __name__ == "__main__":
    main()
