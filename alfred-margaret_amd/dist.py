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


def split_single_haystack(text, world, max_needle_code_points):
    """ONE haystack on `world` GPUs: rank r owns the end positions in (lo, hi] and scans
    text[start:scan_hi], where start = lo minus an overlap of one maximal match, moved back to a code-point
    boundary, and scan_hi = hi moved FORWARD to the next code-point boundary (at most 3 bytes): a slice that
    ends inside a code point would hand the general kernel a truncated sequence, which its guarded decode
    (am_image.h ac_scan_unit) reads as a different, bogus code point.  Whether a needle ends at a position
    depends only on the max-needle-length bytes before it, so a scan started `overlap` earlier reports exactly
    the reference's matches in (lo, hi].  The overlap is 4 bytes per needle code point: under IgnoreCase a
    haystack code point may be longer than the needle code point it lowers to (K, 3 bytes, lowers to k).
    Returns [(start, lo, hi, scan_hi)] per rank."""
    text = memoryview(text)
    n = len(text)
    overlap = 4 * max(int(max_needle_code_points), 1)
    out = []
    for r in range(world):
        lo, hi = shard_bounds(n, r, world)
        start = max(0, lo - overlap)
        while start > 0 and (text[start] & 0xC0) == 0x80:
            start -= 1
        scan_hi = hi
        while scan_hi < n and (text[scan_hi] & 0xC0) == 0x80:
            scan_hi += 1
        out.append((start, lo, hi, scan_hi))
    return out


def own_records(records, start, lo, hi):
    """Filter + rebase the records of a scan of text[start:scan_hi] to the rank's own range: keeps end
    positions in (lo, hi] and makes them relative to the whole haystack."""
    end = records["end_pos"] + start
    keep = (end > lo) & (end <= hi)
    out = records[keep].copy()
    out["end_pos"] = end[keep]
    return out


def gather_records(local_records, first_haystack, dst=0):
    """match lists are not gathered on the device -- every rank copies its own records to the host and rank
    `dst` concatenates them in haystack order.  local_records: numpy array of am_match records of this rank's block
    (haystack ids relative to the block); first_haystack: global index of the block's first haystack.  Returns the global
    record array on rank dst (None elsewhere); with contiguous blocks (shard_bounds) it is sorted by (haystack, end_pos)."""
    import numpy as np
    recs = local_records.copy()
    recs["haystack"] = recs["haystack"] + np.uint32(first_haystack)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return recs
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(recs, parts, dst=dst)
    return np.concatenate(parts) if parts is not None else None
