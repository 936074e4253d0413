import sys, time, torch
sys.path.insert(0, "/root/repo")
from diral_amd.config import bench_config
from diral_amd.vec_env import VecV2VEnv
for name, N, A, L, B, kw in (("c2 general", 64, 32, 2000.0, 4096, {}), ("c2 PF", 64, 32, 2000.0, 4096, dict(proportional_fair=True)),
                             ("c2 posdist type1", 64, 32, 2000.0, 4096, dict(State=dict(add_positional_dist_type=1))),
                             ("c5 general", 128, 64, 4000.0, 4096, {}), ("c3 general", 256, 64, 4000.0, 2048, {})):
    cfg = bench_config(N, A, L, **kw)
    env = VecV2VEnv(cfg, batch=B)
    env.reset_topology(seed=1)
    if "general" in name:
        env.force_general_kernel(True)
    acts = [env.sample(seed=i) for i in range(8)]
    for t in range(80):
        env.step(acts[t % 8], t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200 if N <= 64 else 40
    for t in range(80, 80 + n):
        env.step(acts[t % 8], t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-18s B=%d  %.1f us/slot  %.3g agent-steps/s  kernel %d" % (name, B, dt * 1e6, B * N / dt, env.last_kernel()))
