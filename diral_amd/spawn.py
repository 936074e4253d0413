"""One process per GPU without typing the torchrun command line.

The reference runs one env per process (main_test.py:46); a sharded job here is N such
processes, one per GPU of the node, rendezvousing over 127.0.0.1.  `spawn_ranks` re-executes a
script under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` with a free
port, streams the children's stdout/stderr through and returns their exit status, so that
``python bench.py --gpus 8`` works when it is started the way a single-GPU run is started.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence


def free_port() -> int:
    """A TCP port nobody listens on right now (bound to 127.0.0.1, closed again)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def under_launcher(environ=None) -> bool:
    """True when this process was started by torch.distributed.run (or any launcher that
    exports the rendezvous environment)."""
    env = os.environ if environ is None else environ
    return "WORLD_SIZE" in env and "RANK" in env


def torchrun_command(script: str, argv: Sequence[str], nproc: int, port: Optional[int] = None) -> List[str]:
    if nproc < 1:
        raise ValueError("nproc must be >= 1, got %d" % nproc)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
            "--master-addr", "127.0.0.1", "--master-port", str(port if port else free_port()), script] + list(argv)


def spawn_ranks(script: str, argv: Sequence[str], nproc: int, env=None, port: Optional[int] = None,
                timeout: Optional[float] = None, stdout=None, stderr=None) -> int:
    """Run `script argv` as `nproc` ranks on this node; returns the launcher's exit status
    (non-zero as soon as any rank fails: torch.distributed.run tears the others down)."""
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on these hosts (RCCL across processes)
    e.setdefault("OMP_NUM_THREADS", "1")
    cmd = torchrun_command(script, argv, nproc, port)
    # a session of its own: the launcher AND its rank processes form one process group that a timeout can take down as
    # a whole (the elastic agent cannot tear its workers down when it is SIGKILLed itself - they would keep the GPUs
    # and the rendezvous port)
    proc = subprocess.Popen(cmd, env=e, stdout=stdout, stderr=stderr, start_new_session=True)

    def stop(first_signal):
        import signal
        try:
            os.killpg(proc.pid, first_signal)              # the launcher forwards it and joins its workers
        except ProcessLookupError:
            return
        try:
            proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass
        try:
            os.killpg(proc.pid, signal.SIGKILL)            # whoever is left of the group
        except ProcessLookupError:
            pass
        proc.wait()

    try:
        return int(proc.wait(timeout=timeout))
    except subprocess.TimeoutExpired:
        import signal
        stop(signal.SIGTERM)
        return 124
    except KeyboardInterrupt:
        import signal
        stop(signal.SIGINT)
        raise


def check_visible_gpus(wanted: int, visible: int) -> Optional[str]:
    """None if `wanted` ranks fit the node, else the message to fail with."""
    if wanted < 1:
        return "--gpus must be >= 1, got %d" % wanted
    if visible < wanted:
        return "%d GPUs needed, %d visible" % (wanted, visible)
    return None
