"""Starting the rank processes when no launcher did (python bench.py --gpus N)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

from .common import ROOT

def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port

def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters: this process has not touched
    the GPU and never does), one per GPU, wait, relay rank 0's JSON line.  Any rank failing fails the run."""
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": env0.get("MASTER_PORT", str(free_port())),
                 "BENCH_SPAWNED": "1", "HSA_ENABLE_IPC_MODE_LEGACY": env0.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    line_file = tempfile.TemporaryFile()  # rank 0's stdout: the one JSON line
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py")] + sys.argv[1:], env=env,
                                      stdout=line_file if r == 0 else sys.stderr.fileno()))
    # wait for all of them; the first rank that fails ends the run (its peers would otherwise sit in a collective until the
    # process group's own timeout)
    deadline = time.time() + float(os.environ.get("BENCH_RANK_TIMEOUT_S", "900"))
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        failed_now = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if failed_now or time.time() > deadline:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()  # exactly the processes started here
                    codes[r] = p.wait()
                    if not failed_now:
                        print(f"bench.py: rank {r} did not finish in time", file=sys.stderr)
            break
        time.sleep(0.05)
    rc = 0
    for r, code in enumerate(codes):
        if code != 0:
            print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
            rc = rc or (code if code and code > 0 else 1)
    line_file.seek(0)
    line = line_file.read()
    if rc == 0 and not line.strip():
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    return rc
