"""Multi-GPU harness for the operator bench: one process per GPU, batch sharded across ranks
(the operator is independent per (batch, head)), no data-path collective.  The only
exchange a data-parallel step has on this path is the all-reduce of the mixing-weight gradient dW
(an ordinary parameter gradient that DDP would all-reduce) -- issued here with torch.distributed
("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import os
import signal
import socket
import subprocess
import sys
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world).  Initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # MHLA_DIST_BACKEND=gloo: test the harness with several ranks sharing one GPU
            backend = os.environ.get("MHLA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def launched_by_rendezvous() -> bool:
    """True when the process was started as one rank of a job (torch.distributed.run or spawn_local_ranks)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_local_ranks(n: int, argv: Sequence[str], env_extra: Optional[Dict[str, str]] = None,
                      timeout: Optional[float] = None) -> int:
    """Start `argv` n times as ranks 0..n-1 of one single-node job (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in the environment, rendezvous on 127.0.0.1) and wait for them.  What `bench.py --gpus N` does when it
    was not started by torch.distributed.run.  The caller must not have initialised the GPU: every rank is a fresh
    child process (never an exec of this one), which inherits stdout / stderr.  Returns 0 when every rank exits 0;
    otherwise the first failing rank's code, after ending the remaining ranks by their own PIDs."""
    if n < 1:
        raise ValueError("n must be >= 1")
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    env.update(env_extra or {})
    procs: List[subprocess.Popen] = []
    for r in range(n):
        procs.append(subprocess.Popen(list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r))))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"[spawn_local_ranks] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                    print(f"[spawn_local_ranks] timeout after {timeout} s", file=sys.stderr)
                break
            time.sleep(0.05)
    finally:
        for r in live:
            if procs[r].poll() is None:
                procs[r].send_signal(signal.SIGTERM)
        for r in live:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc


def shard_batch(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of this rank's samples; remainders go to the first ranks."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_mean_(t: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks (what DDP does to parameter gradients)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(dist.get_world_size())
    return t


class OverlappedGradAllReduce:
    """Mean all-reduce of a parameter gradient the way DDP's reducer schedules it: issued as soon as the gradient exists
    (asynchronously, on the process group's own stream) and waited for only where the optimizer would consume it -- here
    the start of the next step, or the device synchronisation that closes the timed region.  Keeps the tensor alive."""

    def __init__(self):
        self._pending = None

    def issue(self, t: torch.Tensor, prescaled: bool = False) -> None:
        """`prescaled`: t already holds gradient / world_size (e.g. divided inside a captured graph)."""
        self.wait()
        if dist.is_initialized() and dist.get_world_size() > 1:
            if not prescaled:
                t.div_(dist.get_world_size())
            self._pending = (dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t)

    def wait(self) -> None:
        if self._pending is not None:
            self._pending[0].wait()
            self._pending = None


def run_steps(step: Callable[[], None], n: int, step_group: Optional[Callable[[], None]] = None, group: int = 1) -> None:
    """Exactly n steps: n // group calls of `step_group` (one call = `group` steps, e.g. the replay of a graph that holds
    `group` captured steps) and the remainder as single steps."""
    if step_group is not None and group > 1:
        for _ in range(n // group):
            step_group()
        n %= group
    for _ in range(n):
        step()


def timed_steps(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None],
                step_group: Optional[Callable[[], None]] = None, group: int = 1) -> float:
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both sides.
    Returns the MAX over ranks of the elapsed seconds."""
    run_steps(step, warmup, step_group, group)
    sync()
    if dist.is_initialized():
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps(step, steps, step_group, group)
    sync()
    if dist.is_initialized():
        dist.barrier()
    sync()
    el = time.perf_counter() - t0
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = "cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el
