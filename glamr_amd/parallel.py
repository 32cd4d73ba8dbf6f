"""Multi-GPU partitioning of the hot path: sequences (and the persons inside them) are independent units
(global_recon/run_dataset.py:67-105 loops them serially), so a node is used by giving every rank its own contiguous block of
sequences -- no data-path collective.  torch.distributed (RCCL on MI355X, gloo in the CPU tests) is used only to rendezvous,
to agree on the elapsed time (max over ranks) and to gather small per-sequence results on rank 0."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def shard_range(n_items, rank, world):
    """Contiguous, balanced block [lo, hi) of rank `rank`; the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def weak_scaling_seeds(per_rank, rank):
    """bench.py: every rank works on `per_rank` sequences of its own (weak scaling)."""
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """Elapsed time of the slowest rank (what the driver's clock sees)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    """Units all ranks processed (the numerator of the whole-job throughput)."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_results(local_results, dst=0):
    """Per-sequence result summaries (small python objects) collected on rank `dst` in global sequence order."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(local_results)
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(list(local_results), out, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [x for part in out for x in part]
