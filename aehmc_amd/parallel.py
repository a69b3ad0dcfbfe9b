"""Multi-GPU plumbing: one process per GPU, chains sharded contiguously, no data-path
collective during sampling; the only exchange is the gather of samples/diagnostics to
rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests) -- SURVEY.md 8e.

The reference has no counterpart (single chain, single process)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def _on() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if _on() else 1


def rank() -> int:
    return dist.get_rank() if _on() else 0


def shard_chains(num_chains: int, r: int = None, w: int = None):
    """Contiguous chain range [lo, hi) of rank r: sizes differ by at most one."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(num_chains, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def chain_seeds(base_seed: int, num_chains: int, r: int = None, w: int = None):
    """Per-chain seeds are a function of the GLOBAL chain index, so results do not depend
    on how many GPUs the chains are spread over."""
    lo, hi = shard_chains(num_chains, r, w)
    return [base_seed + c for c in range(lo, hi)]


def _cpu_backend() -> bool:
    return dist.get_backend() == "gloo"


def barrier(device=None):
    if _on():
        if device is not None and torch.device(device).type == "cuda" and not _cpu_backend():
            dist.barrier(device_ids=[torch.device(device).index])
        else:
            dist.barrier()


def gather_samples(x: torch.Tensor, dst: int = 0):
    """The path's one exchange step: gather the per-rank rows [C_r, ...] into
    [sum C_r, ...] (rank order) ON RANK ``dst`` -- every other rank returns None.  Each shard
    crosses xGMI once, straight to the destination (RCCL gather = point-to-point sends), instead
    of the (w-1) x shard traffic per rank of an all-gather.  ``dst=None`` all-gathers."""
    if not _on():
        return x
    w = dist.get_world_size()
    x = x.contiguous()
    if _cpu_backend() and x.is_cuda:  # gloo dry runs: stage through the host
        out = gather_samples(x.cpu(), dst)
        return None if out is None else out.to(x.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(w)]
    dist.all_gather(sizes, torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if mx != x.shape[0]:  # ragged shards: pad to the largest
        pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
        x = pad
    if dst is None:
        out = torch.empty((w * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)
        parts = list(out.split(mx, dim=0))
    elif dist.get_rank() == dst:
        out = torch.empty((w * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        parts = list(out.split(mx, dim=0))  # views: the shards land in place
        dist.gather(x, parts, dst=dst)      # (RCCL: grouped point-to-point sends / receives)
    else:
        dist.gather(x, None, dst=dst)
        return None
    if len(set(sizes)) == 1:
        return out
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def shard_checksums(x: torch.Tensor):
    """One int64 per rank: the wrap-around sum of the BIT PATTERNS of the rank's rows -- rank 0 compares them with the
    same sum over each shard of the gathered array (the gather must move the rows bit for bit).  Returns a [world]
    int64 tensor on every rank."""
    own = x.contiguous().view(torch.int64).sum().reshape(1)
    if not _on():
        return own
    if _cpu_backend() and own.is_cuda:
        return shard_checksums_cpu(own.cpu()).to(x.device)
    return shard_checksums_cpu(own)


def shard_checksums_cpu(own: torch.Tensor):
    parts = [torch.zeros_like(own) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, own)
    return torch.cat(parts)


def max_over_ranks(v: float, device=None) -> float:
    if not _on():
        return v
    t = torch.tensor([v], dtype=torch.float64,
                     device="cpu" if (device is None or _cpu_backend()) else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(v: int, device=None) -> int:
    if not _on():
        return v
    t = torch.tensor([v], dtype=torch.int64,
                     device="cpu" if (device is None or _cpu_backend()) else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
