import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
from bench import rollout_sps, rollout_sps_fused, rollout_sps_graph
dev = torch.device("cuda:0")
for f, kw in ((rollout_sps, {}), (rollout_sps_fused, {}), (rollout_sps_fused, {"write_chobs": True}), (rollout_sps_graph, {})):
    r = f(dev, **kw)
    print(f.__name__, kw, "%.2f us/slot  %.3g agent-steps/s  coll %.4f" % (r["ms_per_slot"] * 1e3, r["agent_steps_per_s"], r["collision_fraction"]))
