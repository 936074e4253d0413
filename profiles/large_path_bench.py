"""The three-launch form (csrc/step_large.hpp) timed: on sizes only it runs, and at C3's shape next to the
specialised and the general kernel (what the generality costs).  python profiles/large_path_bench.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from diral_amd.config import bench_config
from diral_amd.vec_env import VecV2VEnv
rows = (("c3 wide", 256, 64, 4000.0, 2048, None), ("c3 general", 256, 64, 4000.0, 2048, "general"), ("c3 large", 256, 64, 4000.0, 2048, "large"),
        ("c2 fast64", 64, 32, 2000.0, 4096, None), ("c2 large", 64, 32, 2000.0, 4096, "large"),
        ("512 / 64", 512, 64, 8000.0, 1024, None), ("1024 / 64", 1024, 64, 16000.0, 256, None),
        ("1024 / 512", 1024, 512, 16000.0, 256, None), ("2048 / 128", 2048, 128, 32000.0, 64, None),
        ("4096 / 128", 4096, 128, 64000.0, 16, None))
for name, N, A, L, B, path in rows:
    env = VecV2VEnv(bench_config(N, A, L), batch=B)
    env.reset_topology(seed=1)
    if path == "general":
        env.force_general_kernel(True)
    if path == "large":
        env.force_large_path()
    acts = [env.sample(seed=i) for i in range(8)]
    for t in range(60):
        env.step(acts[t % 8], t)
    torch.cuda.synchronize()
    n = 100 if N <= 256 else 30
    t0 = time.perf_counter()
    for t in range(60, 60 + n):
        env.step(acts[t % 8], t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-12s N=%-5d A=%-4d B=%-5d %9.1f us/slot  %.3g agent-steps/s  kernel %d  hbm %.2f GB" % (
        name, N, A, B, dt * 1e6, B * N / dt, env.last_kernel(), env.hbm_bytes() / 1e9), flush=True)
    del env
    torch.cuda.empty_cache()
