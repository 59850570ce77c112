"""One command -> N rank processes on one node (what Lightning's strategy='ddp' does for the reference:
src/hardware_utils.py:86-95, src/basemodule.py:229,241).

`launch_ranks(n, script, argv)` starts n FRESH children of `script` with the torchrun environment contract (RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT=<free port>), waits for them and returns the first non-zero
exit status.  The calling process must not have touched the GPU (and this module imports neither torch nor HIP), so it is
safe as the plain-shell entry of bench.py / scripts/run.py; nothing is exec'ed over a process that initialised HIP.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, Optional, Sequence

__all__ = ["launch_ranks", "under_launcher"]


def under_launcher() -> bool:
    """True when this process already is a rank (torchrun or launch_ranks set the environment)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def _free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, script: str, argv: Sequence[str], extra_env: Optional[Dict[str, str]] = None,
                 poll_s: float = 0.05) -> int:
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script), *argv], env=env))
    rc = 0
    pending = set(range(n))
    try:
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[launch] rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
            if pending:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return rc
