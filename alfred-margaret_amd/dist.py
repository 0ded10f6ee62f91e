"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo"
in the CPU tests).  The path shards trivially -- haystacks are independent and the automaton is
read-only -- so there is no collective on the data path:
  1. rank 0 flattens the automaton; the position-independent image is broadcast once,
  2. every rank scans its contiguous block of haystacks,
  3. match counts are summed with one all-reduce (record lists stay with their rank; the host
     concatenates them in haystack order).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of n_items for `rank`; blocks differ by at most one item."""
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def broadcast_image(image, device, src=0):
    """image: uint8 tensor on `device` on rank `src` (anything, e.g. None, elsewhere).
    Returns the image tensor on every rank.  Two collectives: the size, then the bytes."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return image
    rank = dist.get_rank()
    size = torch.tensor([image.numel() if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src)
    if rank != src:
        image = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(image, src)
    return image


def allreduce_sum(values, device):
    """Sum a short list of integers over all ranks (final gather of match counts)."""
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def allreduce_max(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
