"""
Multi-GPU sharding of the hot path: parameter tensors are independent units,
so ranks take disjoint sets of tensors with NO data-path collective; only the packed small-artifact
buffers (ranks, sigma, coefficients, codes: KB..MB) are exchanged at the end.  The fp16 bases stay
resident on their owning GPU (gathering them costs more than computing them: SURVEY.md 8e).
One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests).
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def partition_lpt(rows: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first: sort tensors by size (descending, index as tie-break) and give
    each to the currently lightest rank.  Sizes are extremely skewed (4 shapes hold 97-99 % of a
    ViT), so plain round-robin over names would be unbalanced; tensors are never split."""
    order = sorted(range(len(rows)), key=lambda i: (-int(rows[i]), i))
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += int(rows[i])
    for lst in out:
        lst.sort()
    return out


def gather_small(small: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All ranks contribute their packed small-artifact byte buffer; every rank gets the list
    (padded to the longest, then trimmed).  One all_gather of a few hundred KB."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([small.numel()], dtype=torch.int64, device=small.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    mx = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(mx, dtype=torch.uint8, device=small.device)
    padded[:small.numel()] = small
    outs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(outs, padded, group=group)
    return [o[:int(s.item())] for o, s in zip(outs, sizes)]


def all_reduce_gram(gram: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank N x N task Grams (fp64) in place: the only collective of cluster weighting
    (SURVEY.md 8e-i).  Each rank's Gram covers the parameters it owns; the sum is the Gram of the
    concatenated task vectors, from which clustering.cluster_from_gram derives the labels on every rank."""
    import torch.distributed as dist
    dist.all_reduce(gram, op=dist.ReduceOp.SUM, group=group)
    return gram
