import time, torch, sys
sys.path.insert(0, '.')
from diral_amd import c2_config
from diral_amd.vec_env import VecV2VEnv
dev = torch.device('cuda:0')
env = VecV2VEnv(c2_config(), batch=4096, device=dev, out_dtype=torch.float32)
env.reset_topology(seed=1)
acts = [env.sample(seed=i) for i in range(32)]
mode = env.step_mode
for t in range(300): env._step(mode, acts[t % 32], t, want_chobs=True)
torch.cuda.synchronize()
for rep in range(5):
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a = time.perf_counter()
    ev0.record()
    b = time.perf_counter()
    for t in range(20): env._step(mode, acts[t % 32], t, want_chobs=True)
    c = time.perf_counter()
    ev1.record()
    d = time.perf_counter()
    n = 0
    while not ev1.query(): n += 1
    e = time.perf_counter()
    torch.cuda.synchronize()
    f = time.perf_counter()
    print('ev0.record %.1f us | 20 launches %.1f us | ev1.record %.1f | poll %.1f us (%d queries) | sync %.1f us | wall %.1f | events %.1f' % (
        (b-a)*1e6, (c-b)*1e6, (d-c)*1e6, (e-d)*1e6, n, (f-e)*1e6, (f-a)*1e6, ev0.elapsed_time(ev1)*1e3))
