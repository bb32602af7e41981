"""Rank bookkeeping for multi-GPU runs (one process per GPU, torch.distributed over RCCL/gloo).

The SLS path shards over *independent systems* (and, inside one system, over independent subtrees of
the assembly tree -- SURVEY.md section 8e); there is no data-path collective in either case, so all
this module needs is the unit partition and the barrier / max-over-ranks reductions bench.py uses.
"""


def shard_units(nunits, world, rank):
    """Contiguous, balanced partition of `nunits` independent units over `world` ranks."""
    base, extra = divmod(nunits, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value, world):
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, world):
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
