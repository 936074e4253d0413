import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diral_amd.config import bench_config, c2_config, STEP_MY_STEP
from diral_amd.vec_env import VecV2VEnv
from oracle.oracle import Oracle, SQ_IEEE
N, A, L, Rc = [int(x) for x in sys.argv[1:4]] + [float(sys.argv[4])] if len(sys.argv) > 4 else (8, 3, 300, 250.0)
cfg = bench_config(N, A, float(L), communication_range=Rc)
B = 2
rng = np.random.default_rng(1)
x0 = rng.integers(0, int(L), size=(B, N)).astype(np.float64)
v0 = rng.uniform(1.1, 2.7, size=(B, N))
env = VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=torch.float64)
env.reset_topology(x0, 0.0, v0)
orc = Oracle(cfg, batch=B, sq_mode=SQ_IEEE)
orc.reset(x0, np.zeros((B, N)), v0)
for t in range(40):
    a = rng.integers(0, A, size=(B, N)).astype(np.int32)
    obs, rew, done = env.step(a, t)
    o_rew, o_chobs = orc.step(STEP_MY_STEP, a, t)
    o_state = orc.obtain_state(a, o_chobs, o_rew)
    st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
    oe = orc.export()
    bad = []
    if not np.array_equal(obs.cpu().numpy(), o_state): bad.append("state")
    for k in ("seq", "x", "pos_x"):
        if not np.array_equal(st[k], oe[k]): bad.append(k)
    if not np.array_equal(st["age"], np.minimum(oe["age"], 255)): bad.append("age")
    print(t, "kernel", env.last_kernel(), "bad:", bad)
    if bad:
        for k in bad:
            if k == "state":
                idx = np.argwhere(obs.cpu().numpy() != o_state)[:6]; print(" state idx", idx.tolist())
                e, u = idx[0][0], idx[0][1]
                print(" gpu state", obs.cpu().numpy()[e, u]); print(" orc state", o_state[e, u])
                print(" x entries", oe["x"][e, u], "age", oe["age"][e, u], "seq", oe["seq"][e, u], "own pos", oe["pos_x"][e, u])
                print(" v =", oe["x"][e, u] - oe["pos_x"][e, u])
                print(" gpu x entries", st["x"][e, u], "gpu age", st["age"][e, u])
                import ctypes
                buf = (ctypes.c_ulonglong * (B * 4096))()
                fn = env.lib.diral_env_debug_timing
                fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
                print(" dbg rc", fn(env._h, buf, 512))
                arr = np.frombuffer(buf, dtype=np.float64).reshape(B, 64, 64)
                print(" xg used by viewer %d (per subject k):" % u, arr[e, :N, u], "nonzero", np.count_nonzero(arr))
                print(arr[e, :N, :N])
            else:
                g = st[k]; o = np.minimum(oe[k], 255) if k == "age" else oe[k]
                idx = np.argwhere(g != o)[:8]
                print(" ", k, [(tuple(i), g[tuple(i)], o[tuple(i)]) for i in idx])
        print("seq gpu env0\n", st["seq"][0], "\nseq orc\n", oe["seq"][0])
        print("age gpu env0\n", st["age"][0], "\nage orc\n", oe["age"][0])
        break
