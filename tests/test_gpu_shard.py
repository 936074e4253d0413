"""SURVEY 8e on the HIP path on a ONE-GPU box: two processes (torch.distributed.run, gloo group over 127.0.0.1) each step
their `make_sharded_env` half of an 8192-env C2 batch on cuda:0 for 30 slots; `gather_metrics` (the job's only
collective: one all-reduce of seven doubles) must return what ONE handle holding all 8192 envs reports, and the halves'
final states must be the halves of that handle's state.  Keeps `bench.py --gpus N` honest until a multi-GPU node
exists; no scaling figure is taken here (main_test.py:46: one env per process in the reference)."""
import json
import os
import tempfile

import pytest
import torch

from diral_amd import spawn
from diral_amd.config import KERNEL_FAST64, c2_config
from diral_amd.metrics import gather_metrics
from diral_amd.shard import env_shard

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "helpers", "shard_probe.py")


@pytest.mark.parametrize("world,envs,slots", [(2, 8192, 30), (3, 1000, 12)])
def test_sharded_ranks_on_one_gpu_equal_the_single_handle(world, envs, slots):
    from diral_amd.vec_env import VecV2VEnv
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        rc = spawn.spawn_ranks(PROBE, ["--envs", str(envs), "--slots", str(slots)], world, env=env, stdout=out, stderr=err,
                               timeout=600)
        out.seek(0)
        err.seek(0)
        lines = [l for l in out.read().splitlines() if l.startswith("{")]
        assert rc == 0, err.read()[-3000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["world"] == world
    # the same batch on one handle
    one = VecV2VEnv(c2_config(track_prr=True), batch=envs, device="cuda:0")
    one.reset_topology(seed=1234)
    for t in range(slots):
        one.step(one.sample(seed=1000 + t), t)
    one.check()
    want = gather_metrics(one)
    got = rec["summary"]
    assert got["envs"] == envs and got["env_slots"] == envs * slots
    for k in ("envs", "env_slots", "sum_reward", "collision_fraction", "mean_reward_per_agent_step", "mean_collision_metric"):
        assert got[k] == want[k], (k, got[k], want[k])               # counts and integer-valued rewards: exact
    assert abs(got["prr"] - want["prr"]) < 1e-12                     # float sums in another order
    st = one.export_state(tables=True)
    for r in rec["ranks"]:
        s, c = env_shard(envs, r["rank"], world)
        assert (r["start"], r["count"]) == (s, c)
        assert r["pos_x_sum"] == float(st["pos_x"][s:s + c].sum().item())
        assert r["seq_sum"] == int(st["seq"][s:s + c].to(torch.int64).sum().item())
        assert r["age_sum"] == int(st["age"][s:s + c].to(torch.int64).sum().item())
        assert (r["kernel"] & 15) == KERNEL_FAST64
