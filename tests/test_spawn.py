"""bench.py's self-launch (diral_amd/spawn.py): `python bench.py --gpus N` without torchrun re-executes itself
as N ranks under torch.distributed.run on 127.0.0.1 with a free port; exit status propagates.  CPU only (gloo)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

from diral_amd import spawn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "helpers", "spawn_probe.py")


def test_free_port_is_bindable_and_command_line():
    p = spawn.free_port()
    assert 1024 <= p < 65536
    with socket.socket() as s:
        s.bind(("127.0.0.1", p))
    cmd = spawn.torchrun_command("bench.py", ["--gpus", "4", "--steps", "5"], 4, port=29777)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "5"]


def test_visible_gpu_check_and_launcher_detection():
    assert spawn.check_visible_gpus(2, 1) == "2 GPUs needed, 1 visible"
    assert spawn.check_visible_gpus(8, 8) is None
    assert spawn.check_visible_gpus(0, 8) is not None
    assert spawn.under_launcher({"RANK": "0", "WORLD_SIZE": "2"})
    assert not spawn.under_launcher({"WORLD_SIZE": "2"})
    assert not spawn.under_launcher({})


def test_spawn_two_ranks_rendezvous_and_one_json_line():
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        # a stale launcher environment of the parent must not leak into the children
        env = dict(os.environ, RANK="7", WORLD_SIZE="9", MASTER_PORT="1")
        rc = spawn.spawn_ranks(PROBE, ["--gpus", "2"], 2, env=env, stdout=out, stderr=err, timeout=300)
        out.seek(0)
        err.seek(0)
        lines = [l for l in out.read().splitlines() if l.startswith("{")]
        assert rc == 0, err.read()[-2000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["world"] == 2 and rec["sum"] == 3.0 and rec["master"] == "127.0.0.1" and rec["gpus"] == 2
    assert rec["port"] != 1


def test_spawn_propagates_a_failing_rank():
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        rc = spawn.spawn_ranks(PROBE, ["--fail-rank", "1"], 2, stdout=out, stderr=err, timeout=300)
    assert rc != 0


def test_bench_gpus_2_without_gpus_fails_with_a_clear_message():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2
    assert "2 GPUs needed, 0 visible" in r.stderr
    assert "torch.distributed.run" not in r.stderr
