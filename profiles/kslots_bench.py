import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda:0')
for name, fn in (("fused 1/launch", lambda: bench.rollout_sps_fused(dev)),
                 ("K=25", lambda: bench.rollout_sps_kslots(dev)),
                 ("K=25 no per-slot outputs", lambda: bench.rollout_sps_kslots(dev, per_slot_outputs=False)),
                 ("K=5 + state", lambda: bench.rollout_sps_kslots(dev, K=5, launches=80, warm=16, want_obs=True)),
                 ("K=100", lambda: bench.rollout_sps_kslots(dev, K=100, launches=4, warm=1))):
    r = fn()
    print("%-28s %.2f us/slot  %.3f G agent-steps/s  coll %.4f" % (name, r["ms_per_slot"] * 1e3, r["agent_steps_per_s"] / 1e9, r["collision_fraction"]))
