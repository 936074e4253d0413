import sys, torch, numpy as np
sys.path.insert(0, ".")
from diral_amd import c2_config
from diral_amd.vec_env import VecV2VEnv
from diral_amd.sps import SpsPolicy
cfg = c2_config()
B, N = 64, 64
env = VecV2VEnv(cfg, batch=B)
env.reset_topology(seed=1234)
pol = SpsPolicy(B, N, 32, seed=0)
acts = pol.prev_action.clone()
for t in range(400):
    env._step(0, acts, t, want_chobs=True)
    acts = pol.step_from_chobs(env._chobs, acts)
    if t % 25 == 24: env.update_velocity(seed=t)
st = env.export_state()
seq = st["seq"].cpu().numpy().astype(np.int64)   # [B][viewer][subject]
own = np.stack([np.diagonal(seq[b]) for b in range(B)])     # [B][N]
lag = own[:, None, :] - seq
l = lag[seq != 0]
h = np.bincount(np.minimum(l, 64))
c = np.cumsum(h) / h.sum()
print("SPS: max lag", int(l.max()), "cum:", " ".join("%d:%.4f" % (i, c[i]) for i in (0,1,2,3,5,7,11,15,23,31,47,63) if i < len(c)))
for Lm in (7, 15, 31, 63):
    ok = []
    for b in range(B):
        m = np.where(seq[b] != 0, lag[b], 0).max(axis=0)      # per subject
        ok.append((m.reshape(-1, 16).max(1) <= Lm).mean())
    print("  16-subject groups with max lag <= %d: %.4f" % (Lm, float(np.mean(ok))))
    okc = (np.where(seq != 0, lag, 0).max(axis=1) <= Lm).mean()
    print("  single subject columns with max lag <= %d: %.4f" % (Lm, float(okc)))
