import os, sys, torch, numpy as np
sys.path.insert(0, ".")
from diral_amd.config import bench_config
from diral_amd.vec_env import VecV2VEnv
for name,(N,A,L,vary) in {"c3":(256,64,4000.0,False),"c5":(128,64,4000.0,True),"c2":(64,32,2000.0,False)}.items():
    cfg = bench_config(N, A, L, mobility_vary=vary)
    env = VecV2VEnv(cfg, batch=16)
    env.reset_topology(seed=1234)
    sticky = float(os.environ.get("STICKY", "0"))                # probability an agent keeps its resource (converged policy)
    acts = env.sample(seed=999)
    for t in range(160):
        new = env.sample(seed=1000+t)
        if sticky > 0:
            keep = torch.rand((16, N), device=acts.device) < sticky
            acts = torch.where(keep, acts, new)
        else:
            acts = new
        env.step(acts, t)
        if t % 25 == 24: env.update_velocity(seed=t)
    st = env.export_state()
    seq = st["seq"].cpu().numpy().astype(np.int64)   # [B][N(subject k)][N(viewer u)]?
    B = seq.shape[0]
    own = np.stack([np.diagonal(seq[b]) for b in range(B)])     # [B][N]
    # try both orientations: lag must be >= 0
    for orient in (0, 1):
        lag = (own[:, :, None] - seq) if orient == 0 else (own[:, None, :] - seq)
        if (lag[seq != 0] >= 0).all():
            break
    l = lag[seq != 0]
    h = np.bincount(np.minimum(l, 40))
    print(name, "orient", orient, "never-heard frac", float((seq == 0).mean()), "max lag", int(l.max()))
    c = np.cumsum(h) / h.sum()
    print("  cum frac lag<=x:", " ".join("%d:%.5f" % (i, c[i]) for i in (0,1,2,3,4,5,6,7,8,11,15,23,31) if i < len(c)))
    # per (env, 32-column group): fraction of groups whose max lag <= 7 / <= 15
    for Lm in (7, 15, 31):
        ok = []
        for b in range(B):
            lg = np.where(seq[b] != 0, lag[b], 0)
            m = lg.max(axis=1 if orient == 0 else 0)     # per subject
            ok.append((m.reshape(-1, 16).max(1) <= Lm).mean())
        print("  16-subject groups with max lag <= %d: %.4f" % (Lm, float(np.mean(ok))))
