"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The PCFA path shards in two ways:
  * per-pair attacks: independent image pairs, round-robin over ranks, NO data-path collective;
    only the 13 result floats per pair are gathered on rank 0 at the end (`gather_rows`);
  * universal attack: data parallel over the batch; per closure one all-reduce(AVG) of
    d(loss)/d(delta) (<= 10.8 MB fp32) and of the scalar loss (`allreduce_closure`).  All ranks
    receive identical reduced values, so their L-BFGS states stay bit-identical.
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


def allreduce_closure(params, loss):
    """Average the parameter gradients and the loss over ranks (in place); returns the averaged loss."""
    if not is_dist():
        return loss
    n = world_size()
    for p in params:
        if p.grad is not None:
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
            p.grad.div_(n)
    red = loss.detach().clone()
    dist.all_reduce(red, op=dist.ReduceOp.SUM)
    return red / n


def mean_scalar(value, device):
    """Mean of a python float over ranks."""
    if not is_dist():
        return value
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item() / world_size())


def max_scalar(value, device):
    if not is_dist():
        return value
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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
