"""
Mask operators on the GPU with the reference's names and semantics
(reference src/svd_hybrid/mask_loader.py:412-485 combine, :488-648 combine_masks,
:651-709 apply / complement, :712-763 scatter back).  File loaders are out of scope
(SURVEY.md section 2).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import _native as nat
from .pipeline import prepare_vector, resolve_device, _ptr, _stream_ptr


def _as_mask_bytes(m: torch.Tensor, dev) -> torch.Tensor:
    if m.dtype != torch.bool:
        m = m != 0
    return m.to(dev).contiguous().view(-1).view(torch.uint8)


def _combine(masks: List[torch.Tensor], strategy: str) -> torch.Tensor:
    if not masks:
        raise ValueError("Empty mask list")
    if strategy not in nat.MASK_STRATEGIES:
        raise ValueError(f"Unknown mask strategy: {strategy}")
    shape = masks[0].shape
    for m in masks[1:]:
        if m.shape != shape:
            raise ValueError(f"Shape mismatch: mask {m.shape} vs mask {shape}")
    out_dev = masks[0].device
    dev = resolve_device(out_dev if masks[0].is_cuda else "cuda")
    flat = [_as_mask_bytes(m, dev) for m in masks]
    numel = flat[0].numel()
    if numel == 0:
        return torch.zeros(shape, dtype=torch.bool, device=out_dev)
    lib = nat.lib()
    table = torch.tensor([f.data_ptr() for f in flat], dtype=torch.int64).to(dev)
    out = torch.empty(numel, dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    work = torch.empty(int(lib.svdq_mask_work_bytes(numel)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_mask_combine(_ptr(table), len(flat), numel, nat.MASK_STRATEGIES[strategy], _ptr(out),
                                        _ptr(count), _ptr(work), _stream_ptr()), "svdq_mask_combine")
    return out.view(torch.bool).view(shape).to(out_dev)


def compute_union_mask(masks: List[torch.Tensor]) -> torch.Tensor:
    return _combine(masks, "union")


def compute_intersection_mask(masks: List[torch.Tensor]) -> torch.Tensor:
    return _combine(masks, "intersection")


def compute_majority_mask(masks: List[torch.Tensor], threshold: float = 0.5) -> torch.Tensor:
    if threshold != 0.5:
        raise ValueError("only the reference's default threshold 0.5 is implemented on the HIP path")
    return _combine(masks, "majority")


def combine_masks(task_masks: Dict[str, Dict[str, torch.Tensor]], strategy: str = "union", device: str = "cpu",
                  verbose: bool = True) -> Dict[str, torch.Tensor]:
    """Reference mask_loader.py:488-648: per parameter, combine the masks of the tasks that have it."""
    if not task_masks:
        return {}
    names = set()
    for pm in task_masks.values():
        if pm is not None:
            names.update(pm.keys())
    combined = {}
    for name in names:
        lst = [pm[name].to(device) for pm in task_masks.values() if pm is not None and name in pm]
        if not lst:
            continue
        if strategy not in nat.MASK_STRATEGIES:
            raise ValueError(f"Unknown mask strategy: {strategy}")
        combined[name] = _combine(lst, strategy)
    return combined


def compact(vectors: List[torch.Tensor], mask: torch.Tensor, invert: bool = False):
    """Order-preserving compaction of several same-shape fp32 tensors under one mask, on device.
    Returns (flat device tensors sized for the worst case, device int64 count[1], keep-alive)."""
    lib = nat.lib()
    dev = resolve_device(vectors[0].device if vectors[0].is_cuda else "cuda")
    srcs = [prepare_vector(v, dev) for v in vectors]
    numel = srcs[0].numel()
    mb = _as_mask_bytes(mask, dev)
    dsts = [torch.empty(numel, dtype=torch.float32, device=dev) for _ in srcs]
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    if numel == 0:
        return dsts, count, None
    stab = torch.tensor([s.data_ptr() for s in srcs], dtype=torch.int64).to(dev)
    dtab = torch.tensor([d.data_ptr() for d in dsts], dtype=torch.int64).to(dev)
    work = torch.empty(int(lib.svdq_mask_work_bytes(numel)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_mask_compact(_ptr(stab), _ptr(dtab), len(srcs), _ptr(mb), int(bool(invert)), numel,
                                        _ptr(count), _ptr(work), _stream_ptr()), "svdq_mask_compact")
    return dsts, count, (srcs, stab, dtab, work, mb)


def apply_mask_to_tensor(tensor: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Reference mask_loader.py:651-679: flat[mask]."""
    if tensor.shape != mask.shape:
        raise ValueError(f"Shape mismatch: tensor {tensor.shape} vs mask {mask.shape}")
    return _select(tensor, mask, False)


def get_unmasked_portion(tensor: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Reference mask_loader.py:682-709: flat[~mask]."""
    if tensor.shape != mask.shape:
        raise ValueError(f"Shape mismatch: tensor {tensor.shape} vs mask {mask.shape}")
    return _select(tensor, mask, True)


def _select(tensor: torch.Tensor, mask: torch.Tensor, invert: bool) -> torch.Tensor:
    out_dev = tensor.device
    if tensor.numel() == 0:
        return tensor.flatten()
    if tensor.dtype != torch.float32:
        # integer / half inputs only occur in the reference's docstring examples; exact through fp32
        # for |x| < 2^24 and converted back
        res = _select(tensor.to(torch.float32), mask, invert)
        return res.to(tensor.dtype)
    dsts, count, keep = compact([tensor], mask, invert)
    n = int(count.item())
    return dsts[0][:n].clone().to(out_dev)


def reconstruct_from_masked(masked_values: torch.Tensor, unmasked_values: Optional[torch.Tensor], mask: torch.Tensor,
                            original_shape: torch.Size) -> torch.Tensor:
    """Reference mask_loader.py:712-763: zeros; result[mask] = masked; result[~mask] = unmasked
    (svdq_mask_expand: the inverse of the compaction, same tile scan)."""
    lib = nat.lib()
    out_dev = masked_values.device
    dev = resolve_device(out_dev if masked_values.is_cuda else "cuda")
    numel = mask.numel()
    if numel == 0:
        return torch.zeros(original_shape, dtype=masked_values.dtype, device=out_dev)
    if masked_values.dtype != torch.float32:
        res = reconstruct_from_masked(masked_values.float(), None if unmasked_values is None else
                                      unmasked_values.float(), mask, original_shape)
        return res.to(masked_values.dtype)
    mb = _as_mask_bytes(mask, dev)
    # one spare element so that an all-False / all-True mask never hands the kernel an empty buffer
    sig = torch.cat([prepare_vector(masked_values, dev), torch.zeros(1, device=dev)])
    noi = None if unmasked_values is None else torch.cat([prepare_vector(unmasked_values, dev),
                                                          torch.zeros(1, device=dev)])
    out = torch.empty(numel, dtype=torch.float32, device=dev)
    work = torch.empty(int(lib.svdq_mask_work_bytes(numel)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_mask_expand(_ptr(sig), _ptr(noi), _ptr(mb), numel, _ptr(out), _ptr(work), _stream_ptr()),
                  "svdq_mask_expand")
    return out.view(original_shape).to(out_dev)
