"""
Multi-GPU sharding of the hot path: parameter tensors are independent units,
so ranks take disjoint sets of tensors with NO data-path collective; only the packed small-artifact
buffers (ranks, sigma, coefficients, codes: KB..MB) are exchanged at the end of a step.  The fp16 bases stay
resident on their owning GPU (gathering them costs more than computing them: SURVEY.md 8e); ``BasisGather``
exists so that cost can be measured and reported on its own.
One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

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


class RaggedGather:
    """All-gather of one byte buffer per rank whose lengths differ between ranks but not between steps.

    Plan time (constructor): the lengths are exchanged ONCE and the padded send / receive buffers are allocated.
    Step time (``run``): one device-side copy into the send slot and ONE ``all_gather_into_tensor`` -- no size
    exchange, no allocation, no host synchronisation, so it can sit inside a timed (or graph-captured) region.
    ``views()`` returns rank r's bytes as a slice of the receive buffer (valid after the collective has run on
    the stream).

    ``run(buf, overlap=True)`` is the pipelined form: the collective is launched asynchronously on one of TWO
    send / receive buffer sets and the caller's stream does not wait for it, so the next step's kernels run while the
    small artifacts of this step travel; a set is waited for only when it comes up for reuse two steps later (a
    stream-level wait, the host never blocks).  ``finish()`` makes the caller's stream wait for everything in flight;
    ``views()`` then refers to the most recent step.

    A backend that cannot take device tensors (``gloo``: the CPU rehearsal of the multi-GPU run) is served from the SAME
    call: the send / receive sets then live in pinned host memory, ``run`` stages the device buffer into the send slot
    (one D2H copy + a wait for the caller's stream, the price of the rehearsal backend) and the rest -- alternating sets,
    asynchronous collective, waits on reuse -- is the code path the RCCL run takes."""

    def __init__(self, nbytes_local: int, device, group=None, align: int = 256):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.nbytes = int(nbytes_local)
        device = torch.device(device)
        self._staged = device.type == "cuda" and dist.get_backend(group) != "nccl"
        self._src_device = device
        if self._staged:
            device = torch.device("cpu")
        n = torch.tensor([self.nbytes], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n, group=group)                       # the only size exchange, at plan time
        self.sizes = [int(s.item()) for s in sizes]
        self.stride = (max(self.sizes) + align - 1) // align * align
        self._sets = [(self._buf(max(self.stride, 1), device, zero=True),
                       self._buf(max(self.stride, 1) * self.world, device))]
        self._work = [None]
        self._cur = 0
        self._device = device
        self.send, self.recv = self._sets[0]

    def _buf(self, n: int, device, zero: bool = False) -> torch.Tensor:
        t = (torch.zeros if zero else torch.empty)(n, dtype=torch.uint8, device=device)
        return t.pin_memory() if self._staged else t

    def run(self, buf: torch.Tensor, overlap: bool = False) -> torch.Tensor:
        import torch.distributed as dist
        if buf.numel() != self.nbytes:
            raise ValueError(f"buffer has {buf.numel()} bytes, the gather was planned for {self.nbytes}")
        if overlap:
            if len(self._sets) == 1:                                  # the second set, on first use
                self._sets.append((self._buf(self._sets[0][0].numel(), self._device, zero=True),
                                   self._buf(self._sets[0][1].numel(), self._device)))
                self._work.append(None)
            self._cur = (self._cur + 1) % 2
            if self._work[self._cur] is not None:                     # its previous collective (two steps ago)
                self._work[self._cur].wait()
            self.send, self.recv = self._sets[self._cur]
        self.send[:self.nbytes].copy_(buf.view(torch.uint8).reshape(-1), non_blocking=True)
        if self._staged and buf.is_cuda:
            torch.cuda.current_stream(buf.device).synchronize()       # the host collective reads the staged bytes
        if overlap:
            self._work[self._cur] = dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.recv

    def finish(self):
        """The caller's stream waits for every collective still in flight (no-op without ``overlap``)."""
        for i, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[i] = None

    def views(self) -> List[torch.Tensor]:
        return [self.recv[r * self.stride:r * self.stride + self.sizes[r]] for r in range(self.world)]


def gather_small(small: torch.Tensor, group=None, plan: Optional[RaggedGather] = None) -> List[torch.Tensor]:
    """All ranks contribute their packed small-artifact byte buffer; every rank gets the list.  Pass the
    ``RaggedGather`` built at plan time to keep the step free of size exchanges and allocations; without one it
    is built here (convenience for one-off calls)."""
    rg = plan if plan is not None else RaggedGather(small.numel(), small.device, group)
    rg.run(small)
    return rg.views()


def all_reduce_gram(gram: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank N x N task Grams (fp64) in place: the only collective of cluster weighting
    (SURVEY.md 8e-i).  Each rank's Gram covers the parameters it owns; the sum is the Gram of the
    concatenated task vectors, from which clustering.cluster_from_gram derives the labels on every rank."""
    import torch.distributed as dist
    dist.all_reduce(gram, op=dist.ReduceOp.SUM, group=group)
    return gram
