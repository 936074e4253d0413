"""One shape on the three-launch form, for rocprofv3 --kernel-trace --stats: python profiles/large_path_prof.py N A L B [path]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from diral_amd.config import bench_config
from diral_amd.vec_env import VecV2VEnv
N, A, L, B = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
env = VecV2VEnv(bench_config(N, A, L), batch=B)
env.reset_topology(seed=1)
if len(sys.argv) > 5 and sys.argv[5] == "large":
    env.force_large_path()
acts = [env.sample(seed=i) for i in range(8)]
for t in range(60):
    env.step(acts[t % 8], t)
torch.cuda.synchronize()
