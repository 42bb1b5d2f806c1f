"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The PCFA path shards in two ways:
  * per-pair attacks: independent image pairs, round-robin over ranks, NO data-path collective;
    only the 13 result floats per pair are gathered on rank 0 at the end (`gather_rows`);
  * universal attack: data parallel over the batch; per closure ONE all-reduce of a flat buffer
    holding d(loss)/d(delta) of both perturbations and the scalar loss (<= 10.8 MB + 4 B fp32,
    `FlatReducer` / `allreduce_closure`).  All ranks receive identical reduced values, so their
    L-BFGS states stay bit-identical.
Without an initialised process group every helper degrades to the single-process identity.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if is_dist() else 0


def world_size():
    return dist.get_world_size() if is_dist() else 1


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialise from torchrun's env (RANK / WORLD_SIZE / MASTER_*); no-op for a single process."""
    if is_dist() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank())
    dist.init_process_group(backend=backend)
    return True


def shutdown():
    if is_dist():
        dist.destroy_process_group()


def barrier():
    if is_dist():
        dist.barrier()


class FlatReducer:
    """d(loss)/d(delta) of every parameter and the scalar loss in ONE buffer: one collective per closure.

    `pack(loss)` copies the parameters' gradients and the loss into the buffer (plain device copies: it may run
    inside a captured hipGraph, right behind the backward pass); `reduce()` issues the single all-reduce(SUM),
    scales by 1/world and points every `p.grad` at its slice of the buffer, returning the averaged loss.
    The buffer holds sum(numel) + 1 floats (10.8 MB + 4 B for two 3x440x1024 perturbations)."""

    def __init__(self, params):
        self.params = list(params)
        self.n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.n + 1, dtype=p0.dtype, device=p0.device)
        self.collectives = 0

    def pack(self, loss):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self.flat[self.n:].copy_(loss.detach().reshape(1))

    def reduce(self):
        if is_dist():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.collectives += 1
            self.flat.mul_(1.0 / world_size())
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        return self.flat[self.n]


class BatchSums:
    """`--loss cosim` in the multi-rank universal attack: f_cosim = 1 - (p.t / sqrt(p.p)) * sqrt(t.t) (losses.py:88) is a
    ratio of sums over the WHOLE batch, so the gradient a rank sends into its slice depends on the global sums --
    they have to be known BEFORE the backward pass and cannot ride in the gradient all-reduce that follows it.
    Calling the object all-reduces the three local sums in place (12 bytes, SUM) and returns the number of ranks.
    With cosim a closure therefore costs two collectives (this one + FlatReducer's); aee / mse keep one."""

    def __init__(self):
        self.collectives = 0

    def __call__(self, sums):
        if not is_dist():
            return 1
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        self.collectives += 1
        return world_size()


_REDUCERS = {}


def allreduce_closure(params, loss, reducer=None):
    """Average the parameter gradients and the loss over ranks with ONE all-reduce; returns the averaged loss.
    After the call every p.grad ALIASES a slice of the reducer's flat buffer (the next call overwrites it).  Without an
    explicit `reducer` one is kept per parameter set (identity, sizes, device), so repeated calls reuse one buffer."""
    if not is_dist():
        return loss
    if reducer is None:
        params = list(params)
        key = tuple((id(p), p.numel(), str(p.device)) for p in params)
        reducer = _REDUCERS.get(key)
        if reducer is None or any(a is not b for a, b in zip(reducer.params, params)):
            reducer = _REDUCERS[key] = FlatReducer(params)
    reducer.pack(loss)
    return reducer.reduce()


def mean_scalar(value, device):
    """Mean of a python float over ranks."""
    if not is_dist():
        return value
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item() / world_size())


def mean_scalars(values, device):
    """Means of several python floats over ranks with one collective."""
    if not is_dist():
        return tuple(values)
    t = torch.tensor([float(v) for v in values], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple((t / world_size()).tolist())


def max_scalar(value, device):
    if not is_dist():
        return value
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_scalars(value, device):
    """Every rank's python float, in rank order, on every rank."""
    if not is_dist():
        return [float(value)]
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    out = [torch.zeros_like(t) for _ in range(world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_rows(rows, width, device):
    """All ranks contribute a list of `width`-float rows; rank 0 gets the concatenation, others []."""
    if not is_dist():
        return [tuple(r) for r in rows]
    n = torch.tensor([len(rows)], device=device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(counts, n)
    cap = max(int(c.item()) for c in counts)
    buf = torch.full((max(cap, 1), width), float('nan'), device=device, dtype=torch.float64)
    if rows:
        buf[:len(rows)] = torch.tensor(rows, device=device, dtype=torch.float64)
    bufs = [torch.zeros_like(buf) for _ in range(world_size())]
    dist.all_gather(bufs, buf)
    if rank() != 0:
        return []
    out = []
    for c, b in zip(counts, bufs):
        out.extend(tuple(r) for r in b[:int(c.item())].cpu().tolist())
    return out
