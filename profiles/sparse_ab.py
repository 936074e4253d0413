#!/usr/bin/env python3
"""Step time on SPARSE topologies (entries far beyond the codes' reach: keyed quads / flagged passes dominate), for A/B
between source trees:  python profiles/sparse_ab.py [tree_root]"""
import os, sys
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from diral_amd.config import bench_config
from diral_amd.vec_env import VecV2VEnv
for N, A, L, Rc, B in ((64, 32, 9000.0, 140.0, 4096), (128, 64, 20000.0, 140.0, 2048), (256, 64, 30000.0, 140.0, 1024), (256, 64, 12000.0, 250.0, 1024)):
    cfg = bench_config(N, A, L, communication_range=Rc)
    env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float32)
    env.reset_topology(seed=3)
    acts = [env.sample(seed=i) for i in range(16)]
    for t in range(150):
        env.step(acts[t % 16], t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(150, 250):
        env.step(acts[t % 16], t)
    e1.record(); torch.cuda.synchronize()
    print("%s N=%d A=%d L=%g Rc=%g B=%d: %.1f us per step (kernel code %d)" % (os.path.basename(root), N, A, L, Rc, B, e0.elapsed_time(e1) * 10, env.last_kernel()))
