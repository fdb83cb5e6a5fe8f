"""Element sharding across the GPUs of one node.

compute_and_apply_rhs touches only its own element's slices, so the global element
range is cut into contiguous slabs, one per rank, and no data-path collective is needed.
torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests) is used
only around the kernel: start barrier, max-over-ranks timing, and the optional
reduction of the per-rank norm partial sums that the reference driver prints.
"""
import math

import torch

from .caar import shard_range  # noqa: F401  (re-exported)


def max_over_ranks(values, dist=None, device="cpu"):
    """Elementwise MAX of a list of floats over all ranks (identity without a group)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():  # (also a group of one rank: the collective path, RCCL included, is what runs)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


def gather_over_ranks(values, dist=None, device="cpu"):
    """Every rank's list of floats, in rank order: [[rank 0's values], [rank 1's], ...]."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        return [p.tolist() for p in parts]
    return [t.tolist()]


def gather_slab_norms(per_element_sq, dist=None, device="cpu"):
    """print_results_2norm over a sharded element range.

    `per_element_sq` is this rank's (n_local, 3) tensor of pow(compute_norm(field), 2)
    for v, T, dp3d (what caar_launch_state_norms produces).  The reference adds these in
    element order (compute_and_apply_rhs.cpp:388-390); slabs are contiguous and ordered
    by rank, so gathering the slabs in rank order and summing sequentially reproduces
    the single-process result bit for bit.
    """
    local = per_element_sq.to(dtype=torch.float64, device=device).reshape(-1, 3).contiguous()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        counts[dist.get_rank()] = local.shape[0]
        dist.all_reduce(counts)
        nmax = int(counts.max())
        padded = torch.zeros(nmax, 3, dtype=torch.float64, device=device)
        padded[: local.shape[0]] = local
        parts = [torch.zeros_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)
        local = torch.cat([p[: int(c)] for p, c in zip(parts, counts.tolist())], dim=0)
    s = [0.0, 0.0, 0.0]
    for row in local.cpu().tolist():
        for f in range(3):
            s[f] += row[f]
    return tuple(math.sqrt(x) for x in s)
