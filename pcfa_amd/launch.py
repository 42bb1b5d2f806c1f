"""One process per GPU: start N fresh rank processes of a script from a parent that never touches the GPU.

The reference runs one process and wraps its networks in DataParallel (attack_PCFA.py:344-350); here independent
image pairs shard over ranks (attack_PCFA.py:668-670) and the universal attack is data parallel over the batch
(attack_PCFA.py:469-490), one rank per GPU.  `torch.distributed.run` is one way to start the ranks; `spawn_ranks` is
the other, used when a script is started as a plain `python script.py --gpus N`:

    if args.gpus > 1 and not launch.inside_rank():
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

The parent must call it BEFORE anything initialises HIP (no `torch.cuda.is_available()`, no kernel library load): it
only starts children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for all of them and
returns the first non-zero exit code (terminating the remaining ranks when one fails, so a dead rank cannot leave the
others waiting in a collective).  No process is ever replaced (`exec`): every rank is a child started from a clean
interpreter.  Rank 0 inherits the parent's stdout (the one JSON line of bench.py); the other ranks' stdout goes to
stderr.
"""
import os
import socket
import subprocess
import sys
import time

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def inside_rank():
    """True in a process that torchrun or `spawn_ranks` started as one rank of a job."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), PCFA_SPAWNED_RANK="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
    return env


def spawn_ranks(argv, nprocs, env=None, poll_s=0.2, timeout_s=None):
    """Run `python argv...` as `nprocs` ranks on this node; returns the job's exit code (0 = every rank exited 0)."""
    if nprocs < 1:
        raise ValueError("nprocs must be >= 1")
    if inside_rank():
        raise RuntimeError("spawn_ranks called from inside a rank (RANK/WORLD_SIZE already set)")
    port = free_port()
    procs = []
    for r in range(nprocs):
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=rank_env(r, nprocs, port, env), stdout=out))
    t0 = time.monotonic()
    code = 0
    try:
        pending = set(range(nprocs))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    print("rank %d exited with code %d: stopping the other ranks" % (r, rc), file=sys.stderr)
            if code != 0 or (timeout_s is not None and time.monotonic() - t0 > timeout_s):
                if code == 0:
                    code = 124
                    print("ranks still running after %.0f s: stopping them" % timeout_s, file=sys.stderr)
                break
            if pending:
                time.sleep(poll_s)
    finally:
        for p in procs:           # exactly the processes started here, by handle -- never by pattern
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    return code
