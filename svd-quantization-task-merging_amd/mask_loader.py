"""
Mask operators on the GPU with the reference's names and semantics
(reference src/svd_hybrid/mask_loader.py:412-485 combine, :488-648 combine_masks,
:651-709 apply / complement, :712-763 scatter back).  File loaders are out of scope
(SURVEY.md section 2).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch

from . import _native as nat
from .pipeline import prepare_vector, resolve_device, _ptr, _stream_ptr


def _as_mask_bytes(m: torch.Tensor, dev) -> torch.Tensor:
    if m.dtype != torch.bool:
        m = m != 0
    return m.to(dev).contiguous().view(-1).view(torch.uint8)


class MaskSet:
    """A ragged set of masked parameters processed together (svdq_maskset_*): one tile table, a handful of
    launches for all of them, per-parameter counts kept on the device."""

    def __init__(self, numels, device):
        from ctypes import byref, c_int64, c_void_p
        self.lib = nat.lib()
        self.device = resolve_device(device)
        self.numels = [int(n) for n in numels]
        self.Q = len(self.numels)
        self._h = c_void_p()
        arr = (c_int64 * self.Q)(*self.numels)
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_create(byref(self._h), self.Q, arr), "svdq_maskset_create")
        self.work = torch.empty(int(self.lib.svdq_maskset_work_bytes(self._h)), dtype=torch.uint8, device=self.device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            from ctypes import c_void_p
            self.lib.svdq_maskset_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _table(self, tensors):
        return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64).to(self.device)

    # prepare_* allocate outputs and upload pointer tables once; run_* only enqueue kernels, so a caller
    # that compresses the same model repeatedly (or a benchmark) pays no host work per call.
    def prepare_combine(self, masks_per_param, strategy: str):
        if strategy not in nat.MASK_STRATEGIES:
            raise ValueError(f"Unknown mask strategy: {strategy}")
        n = len(masks_per_param[0])
        if n < 1:
            raise ValueError("Empty mask list")
        flat = [[_as_mask_bytes(m, self.device) for m in ms] for ms in masks_per_param]
        for q, ms in enumerate(flat):
            if len(ms) != n or any(m.numel() != self.numels[q] for m in ms):
                raise ValueError("Shape mismatch: every parameter needs the same number of same-shaped masks")
        outs = [torch.empty(nq, dtype=torch.uint8, device=self.device) for nq in self.numels]
        counts = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        self._c = dict(flat=flat, outs=outs, counts=counts, n=n, strategy=nat.MASK_STRATEGIES[strategy],
                       mt=self._table([m for ms in flat for m in ms]), ot=self._table(outs))
        return outs, counts

    def run_combine(self):
        c = self._c
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_combine(self._h, _ptr(c["mt"]), c["n"], c["strategy"], _ptr(c["ot"]),
                                                    _ptr(c["counts"]), _stream_ptr()), "svdq_maskset_combine")

    def combine(self, masks_per_param, strategy: str):
        """masks_per_param[q] = list of n per-task bool masks (same n for every q).
        Returns (combined uint8 byte tensors [numel_q], device int64 counts [Q])."""
        outs, counts = self.prepare_combine(masks_per_param, strategy)
        self.run_combine()
        return outs, counts

    def prepare_compact(self, masks, srcs_per_param, want_false: bool):
        n_src = len(srcs_per_param[0])
        mb = [_as_mask_bytes(m, self.device) for m in masks]
        srcs = [[prepare_vector(v, self.device) for v in vs] for vs in srcs_per_param]
        for q in range(self.Q):
            if mb[q].numel() != self.numels[q] or len(srcs[q]) != n_src or any(v.numel() != self.numels[q] for v in srcs[q]):
                raise ValueError(f"Shape mismatch: tensor vs mask for parameter {q}")
        dt = [[torch.empty(self.numels[q], dtype=torch.float32, device=self.device) for _ in range(n_src)]
              for q in range(self.Q)]
        df = [[torch.empty(self.numels[q], dtype=torch.float32, device=self.device) for _ in range(n_src)]
              for q in range(self.Q)] if want_false else None
        ct = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        cf = torch.zeros(self.Q, dtype=torch.int64, device=self.device) if want_false else None
        self._x = dict(mb=mb, srcs=srcs, n_src=n_src, ct=ct, cf=cf, mt=self._table(mb),
                       st=self._table([v for vs in srcs for v in vs]), tt=self._table([v for vs in dt for v in vs]),
                       ft=self._table([v for vs in df for v in vs]) if want_false else None)
        return dt, df, ct, cf

    def run_compact(self):
        x = self._x
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_compact(self._h, _ptr(x["mt"]), _ptr(x["st"]), _ptr(x["tt"]), _ptr(x["ft"]),
                                                    x["n_src"], _ptr(x["ct"]), _ptr(x["cf"]), _ptr(self.work),
                                                    _stream_ptr()), "svdq_maskset_compact")

    def prepare_indices(self, masks, want_false: bool):
        mb = [_as_mask_bytes(m, self.device) for m in masks]
        for q in range(self.Q):
            if mb[q].numel() != self.numels[q]:
                raise ValueError(f"Shape mismatch: tensor vs mask for parameter {q}")
        it = [torch.empty(self.numels[q], dtype=torch.int32, device=self.device) for q in range(self.Q)]
        if_ = [torch.empty(self.numels[q], dtype=torch.int32, device=self.device) for q in range(self.Q)] \
            if want_false else None
        ct = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        cf = torch.zeros(self.Q, dtype=torch.int64, device=self.device) if want_false else None
        self._i = dict(mb=mb, it=it, if_=if_, ct=ct, cf=cf, mt=self._table(mb), tt=self._table(it),
                       ft=self._table(if_) if want_false else None)
        return it, if_, ct, cf

    def run_indices(self):
        x = self._i
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_indices(self._h, _ptr(x["mt"]), _ptr(x["tt"]), _ptr(x["ft"]), _ptr(x["ct"]),
                                                    _ptr(x["cf"]), _ptr(self.work), _stream_ptr()),
                      "svdq_maskset_indices")

    def prepare_combine_indices(self, masks_per_param, strategy: str, want_false: bool):
        """combine + index build on the combined masks in 3 launches (the combine pass supplies the tile counts)."""
        outs, _ = self.prepare_combine(masks_per_param, strategy)
        it, if_, ct, cf = self.prepare_indices([o.view(torch.bool) for o in outs], want_false)
        return outs, it, if_, ct, cf

    def run_combine_indices(self):
        c, x = self._c, self._i
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_combine_indices(self._h, _ptr(c["mt"]), c["n"], c["strategy"], _ptr(c["ot"]),
                                                            _ptr(x["tt"]), _ptr(x["ft"]), _ptr(x["ct"]), _ptr(x["cf"]),
                                                            _ptr(self.work), _stream_ptr()),
                      "svdq_maskset_combine_indices")

    def prepare_combine_packed_indices(self, streams, bit_offsets, strategy: str, want_false: bool):
        """Combine + index build straight from BIT-PACKED tall masks (numpy.packbits order, one uint8 stream per
        task over the flattened state dict -- the form TALL_mask files have): parameter q's elements are bits
        ``bit_offsets[q] + e`` of every stream.  Returns (combined bool-byte masks, idx_true, idx_false | None,
        count_true, count_false | None); run with run_combine_packed_indices()."""
        if strategy not in nat.MASK_STRATEGIES:
            raise ValueError(f"Unknown mask strategy: {strategy}")
        if not streams:
            raise ValueError("Empty mask list")
        st = [s.to(self.device).contiguous().view(torch.uint8) for s in streams]
        need = max((int(o) + n + 7) // 8 for o, n in zip(bit_offsets, self.numels))
        if any(s.numel() < need for s in st):
            raise ValueError("Shape mismatch: a packed mask stream is shorter than the parameters it should cover")
        outs = [torch.empty(nq, dtype=torch.uint8, device=self.device) for nq in self.numels]
        it = [torch.empty(nq, dtype=torch.int32, device=self.device) for nq in self.numels]
        if_ = [torch.empty(nq, dtype=torch.int32, device=self.device) for nq in self.numels] if want_false else None
        ct = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        cf = torch.zeros(self.Q, dtype=torch.int64, device=self.device) if want_false else None
        self._p = dict(st=st, outs=outs, it=it, if_=if_, ct=ct, cf=cf, n=len(st), strategy=nat.MASK_STRATEGIES[strategy],
                       sp=self._table(st), sb=torch.tensor([s.numel() for s in st], dtype=torch.int64).to(self.device),
                       bo=torch.tensor([int(o) for o in bit_offsets], dtype=torch.int64).to(self.device),
                       ot=self._table(outs), tt=self._table(it), ft=self._table(if_) if want_false else None)
        return outs, it, if_, ct, cf

    def run_combine_packed_indices(self):
        x = self._p
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_combine_packed_indices(
                self._h, _ptr(x["sp"]), _ptr(x["sb"]), _ptr(x["bo"]), x["n"], x["strategy"], _ptr(x["ot"]), _ptr(x["tt"]),
                _ptr(x["ft"]), _ptr(x["ct"]), _ptr(x["cf"]), _ptr(self.work), _stream_ptr()),
                "svdq_maskset_combine_packed_indices")

    # ---- unit starts: what the mask-walk mode of the compressor (CompressPlan.run_masked) needs instead of index lists
    def count_scan(self, masks):
        """mask.sum() / (~mask).sum() of already combined masks (device int64 [Q] each); leaves the tile offsets in
        ``self.work`` for unit_starts()."""
        mb = [_as_mask_bytes(m, self.device) for m in masks]
        for q in range(self.Q):
            if mb[q].numel() != self.numels[q]:
                raise ValueError(f"Shape mismatch: tensor vs mask for parameter {q}")
        ct = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        cf = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        self._s = dict(mb=mb, mt=self._table(mb), ct=ct, cf=cf)
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_count_scan(self._h, _ptr(self._s["mt"]), _ptr(ct), _ptr(cf), _ptr(self.work),
                                                       _stream_ptr()), "svdq_maskset_count_scan")
        return ct, cf

    def unit_starts(self, plan, rows_dev, entry_map=None, mask_table=None) -> torch.Tensor:
        """Source position of every work unit's first row (device int64 [plan units]) for the masks counted last
        (count_scan / run_combine / ...).  ``entry_map``: None (plan parameter p <-> mask p) or a list of
        (mask index, inverted) per plan parameter -- the noise region takes the cleared elements of its mask."""
        mt = mask_table if mask_table is not None else self._s["mt"]
        em = None
        if entry_map is not None:
            # bit 31 = inverted: as a signed int32 that is q - 2^31
            em = torch.tensor([int(q) - (1 << 31) if inv else int(q) for q, inv in entry_map],
                              dtype=torch.int32).to(self.device)
        us = torch.empty(int(plan.sizes.n_units), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_unit_starts(self._h, plan._h, _ptr(mt), _ptr(em), _ptr(rows_dev),
                                                        _ptr(self.work), _ptr(us), _stream_ptr()),
                      "svdq_maskset_unit_starts")
        return us

    def prepare_combine_starts(self, masks_per_param, strategy: str, plan):
        """combine + scan + unit starts in 3 launches (no index lists); run with run_combine_starts().
        Returns (combined uint8 masks, count_true [Q], unit starts [plan units])."""
        outs, _ = self.prepare_combine(masks_per_param, strategy)
        ct = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        cf = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        us = torch.empty(int(plan.sizes.n_units), dtype=torch.int64, device=self.device)
        self._cs = dict(ct=ct, cf=cf, us=us, plan=plan)
        self._s = dict(mb=outs, mt=self._c["ot"], ct=ct, cf=cf)
        return outs, ct, us

    def run_combine_starts(self):
        c, x = self._c, self._cs
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_combine_starts(self._h, x["plan"]._h, _ptr(c["mt"]), c["n"], c["strategy"],
                                                           _ptr(c["ot"]), _ptr(x["ct"]), _ptr(x["cf"]), _ptr(self.work),
                                                           _ptr(x["us"]), _stream_ptr()), "svdq_maskset_combine_starts")

    def prepare_combine_packed_starts(self, streams, bit_offsets, strategy: str, plan):
        """The same from bit-packed tall masks (see prepare_combine_packed_indices)."""
        if strategy not in nat.MASK_STRATEGIES:
            raise ValueError(f"Unknown mask strategy: {strategy}")
        if not streams:
            raise ValueError("Empty mask list")
        st = [s.to(self.device).contiguous().view(torch.uint8) for s in streams]
        need = max((int(o) + n + 7) // 8 for o, n in zip(bit_offsets, self.numels))
        if any(s.numel() < need for s in st):
            raise ValueError("Shape mismatch: a packed mask stream is shorter than the parameters it should cover")
        outs = [torch.empty(nq, dtype=torch.uint8, device=self.device) for nq in self.numels]
        ct = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        cf = torch.zeros(self.Q, dtype=torch.int64, device=self.device)
        us = torch.empty(int(plan.sizes.n_units), dtype=torch.int64, device=self.device)
        self._ps = dict(st=st, outs=outs, ct=ct, cf=cf, us=us, plan=plan, n=len(st),
                        strategy=nat.MASK_STRATEGIES[strategy], sp=self._table(st),
                        sb=torch.tensor([s.numel() for s in st], dtype=torch.int64).to(self.device),
                        bo=torch.tensor([int(o) for o in bit_offsets], dtype=torch.int64).to(self.device),
                        ot=self._table(outs))
        self._s = dict(mb=outs, mt=self._ps["ot"], ct=ct, cf=cf)
        return outs, ct, us

    def run_combine_packed_starts(self):
        x = self._ps
        with torch.cuda.device(self.device):
            nat.check(self.lib.svdq_maskset_combine_packed_starts(
                self._h, x["plan"]._h, _ptr(x["sp"]), _ptr(x["sb"]), _ptr(x["bo"]), x["n"], x["strategy"], _ptr(x["ot"]),
                _ptr(x["ct"]), _ptr(x["cf"]), _ptr(self.work), _ptr(x["us"]), _stream_ptr()),
                "svdq_maskset_combine_packed_starts")

    def indices(self, masks, want_false: bool):
        """Ascending flat positions (int32) of the set / cleared elements of every mask: what the gather mode
        of the compressor reads the task deltas through, instead of 2 N compacted copies per parameter.
        Returns (idx_true[q], idx_false[q] | None, count_true [Q], count_false [Q] | None)."""
        out = self.prepare_indices(masks, want_false)
        self.run_indices()
        return out

    def compact(self, masks, srcs_per_param, want_false: bool):
        """masks[q]: combined mask; srcs_per_param[q]: list of n_src fp32 tensors of that shape.
        Returns (dst_true[q][s], dst_false[q][s] | None, count_true [Q], count_false [Q] | None);
        destination buffers are sized for the worst case, counts stay on the device."""
        out = self.prepare_compact(masks, srcs_per_param, want_false)
        self.run_compact()
        return out


def _combine(masks: List[torch.Tensor], strategy: str, votes_needed: Optional[int] = None) -> torch.Tensor:
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
        code = nat.MASK_STRATEGIES[strategy]
        if votes_needed is not None:
            code |= (int(votes_needed) + 1) << 8
        nat.check(lib.svdq_mask_combine(_ptr(table), len(flat), numel, code, _ptr(out),
                                        _ptr(count), _ptr(work), _stream_ptr()), "svdq_mask_combine")
    return out.view(torch.bool).view(shape).to(out_dev)


def compute_union_mask(masks: List[torch.Tensor]) -> torch.Tensor:
    return _combine(masks, "union")


def compute_intersection_mask(masks: List[torch.Tensor]) -> torch.Tensor:
    return _combine(masks, "intersection")


def compute_majority_mask(masks: List[torch.Tensor], threshold: float = 0.5) -> torch.Tensor:
    """Reference mask_loader.py:456-485: ``vote_sum >= threshold * len(masks)``.  The reference compares an fp32
    count tensor with the Python product; torch casts that scalar to fp32, so the smallest passing count is
    ceil(fp32(threshold * n)) -- computed here on the host and handed to the kernel as an integer."""
    if not masks:
        raise ValueError("Empty mask list")
    if threshold == 0.5:
        return _combine(masks, "majority")
    import math
    import numpy as np
    bound = float(np.float32(threshold * len(masks)))
    need = max(0, min(len(masks) + 1, int(math.ceil(bound)))) if math.isfinite(bound) else (0 if bound < 0 else len(masks) + 1)
    return _combine(masks, "majority", votes_needed=need)


def combine_masks(task_masks: Dict[str, Dict[str, torch.Tensor]], strategy: str = "union", device: str = "cpu",
                  verbose: bool = True) -> Dict[str, torch.Tensor]:
    """Reference mask_loader.py:488-648: per parameter, combine the masks of the tasks that have it
    (tasks whose mask dict is None are skipped).  All parameters that have the same number of masks go
    through ONE batched launch (svdq_maskset_combine)."""
    if not task_masks:
        return {}
    if strategy not in nat.MASK_STRATEGIES:
        raise ValueError(f"Unknown mask strategy: {strategy}")
    names = []
    for pm in task_masks.values():
        if pm is not None:
            for n in pm.keys():
                if n not in names:
                    names.append(n)
    per_param = {n: [pm[n] for pm in task_masks.values() if pm is not None and n in pm] for n in names}
    by_count: Dict[int, List[str]] = {}
    for n, lst in per_param.items():
        if lst and lst[0].numel() > 0:
            by_count.setdefault(len(lst), []).append(n)
    combined: Dict[str, torch.Tensor] = {}
    for cnt, group in by_count.items():
        shape0 = {n: per_param[n][0].shape for n in group}
        for n in group:
            for m in per_param[n][1:]:
                if m.shape != shape0[n]:
                    raise ValueError(f"Shape mismatch: mask {m.shape} vs mask {shape0[n]}")
        dev_in = per_param[group[0]][0].device
        ms = MaskSet([per_param[n][0].numel() for n in group], dev_in if dev_in.type == "cuda" else "cuda")
        outs, _ = ms.combine([per_param[n] for n in group], strategy)
        for n, o in zip(group, outs):
            combined[n] = o.view(torch.bool).view(shape0[n]).to(device)      # the reference moves every mask to `device`
    for n, lst in per_param.items():          # zero-sized parameters
        if lst and lst[0].numel() == 0:
            combined[n] = torch.zeros(lst[0].shape, dtype=torch.bool, device=device)
    if verbose:
        print(f"   combined masks for {len(combined)} parameters with strategy '{strategy}'")
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


# ------------------------------------------------------------------------------- mask files (host plumbing)
def state_dict_to_vector(state_dict: Dict[str, torch.Tensor], remove_keys: Optional[List[str]] = None) -> torch.Tensor:
    """Reference mask_loader.py:66-105: flattened tensors concatenated in sorted-key order."""
    skip = set(remove_keys or [])
    flat = [state_dict[k].flatten() for k in sorted(state_dict.keys()) if k not in skip]
    return torch.cat(flat) if flat else torch.tensor([])


def vector_to_state_dict(vector: torch.Tensor, reference_state_dict: Dict[str, torch.Tensor],
                         remove_keys: Optional[List[str]] = None) -> Dict[str, torch.Tensor]:
    """Inverse of state_dict_to_vector (what mask_loader.py:108-122 intends: its body references an
    undefined name and cannot run as written)."""
    skip = set(remove_keys or [])
    out, pos = {}, 0
    for k in sorted(reference_state_dict.keys()):
        if k in skip:
            continue
        n = reference_state_dict[k].numel()
        out[k] = vector[pos:pos + n].view(reference_state_dict[k].shape)
        pos += n
    return out


def load_tall_mask_file(mask_path: str, reference_state_dict: Dict[str, torch.Tensor],
                        remove_keys: Optional[List[str]] = None, device: str = "cpu") -> Dict[str, Dict[str, torch.Tensor]]:
    """Reference mask_loader.py:125-206: {task: bit-packed mask over the flattened state dict} -> per-parameter
    bool masks.  The file is read with loaders that execute nothing: ``numpy.load`` (an ``.npz`` with one packed
    array per task) or ``torch.load(weights_only=True)`` (a dict of uint8 tensors).  The reference's own files are
    pickled dicts of numpy arrays read with ``weights_only=False``; convert those once with
    ``numpy.savez(path, **packed)``."""
    import numpy as np
    if not os.path.exists(mask_path):
        raise FileNotFoundError(f"TALL mask file not found: {mask_path}")
    packed = None
    try:
        z = np.load(mask_path, allow_pickle=False)
        if hasattr(z, "files"):
            packed = {k: z[k] for k in z.files}
    except Exception:
        packed = None
    if packed is None:
        try:
            packed = torch.load(mask_path, map_location="cpu", weights_only=True)
        except Exception as e:
            raise RuntimeError(f"{mask_path}: not loadable without unpickling arbitrary objects ({e}); "
                               "convert it with numpy.savez(path, **packed_masks)") from e
    skip = set(remove_keys or [])
    expected = sum(v.numel() for k, v in reference_state_dict.items() if k not in skip)
    out = {}
    for task, pm in packed.items():
        arr = pm.cpu().numpy() if isinstance(pm, torch.Tensor) else np.asarray(pm)
        if arr.dtype != np.uint8:
            raise TypeError(f"Unexpected type for packed_mask: {arr.dtype}")
        bits = torch.from_numpy(np.unpackbits(arr)[:expected].copy()).to(device)
        out[task] = {k: v.bool() for k, v in vector_to_state_dict(bits, reference_state_dict, remove_keys).items()}
    return out


def load_single_mask(mask_path: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Reference mask_loader.py:209-239 (safe loader)."""
    if not os.path.exists(mask_path):
        raise FileNotFoundError(f"Mask file not found: {mask_path}")
    masks = torch.load(mask_path, map_location=device, weights_only=True)
    return {k: (m if m.dtype == torch.bool else m.bool()) for k, m in masks.items()}


def load_task_masks(mask_dir: str, task_names: List[str], device: str = "cpu",
                    reference_state_dict: Optional[Dict[str, torch.Tensor]] = None,
                    remove_keys: Optional[List[str]] = None, verbose: bool = True) -> Dict[str, Optional[Dict]]:
    """Reference mask_loader.py:242-409: a TALL_mask_{N}task(s) file if present (and a reference state dict is
    given), else one ``{task}_mask.pt`` / ``{task}.pt`` / ``{task}/mask.pt`` per task; missing -> None."""
    n = len(task_names)
    if reference_state_dict is not None:
        for stem in (f"TALL_mask_{n}task", f"TALL_mask_{n}tasks", f"tall_mask_{n}task", f"tall_mask_{n}tasks"):
            for ext in (".npy", ".npz", ".pt"):
                path = os.path.join(mask_dir, stem + ext)
                if os.path.exists(path):
                    allm = load_tall_mask_file(path, reference_state_dict, remove_keys=remove_keys, device=device)
                    return {t: allm.get(t) for t in task_names}
    out: Dict[str, Optional[Dict]] = {}
    for t in task_names:
        out[t] = None
        for cand in (f"{t}_mask.pt", f"{t}.pt", os.path.join(t, "mask.pt")):
            path = os.path.join(mask_dir, cand)
            if os.path.exists(path):
                out[t] = load_single_mask(path, device)
                break
        if verbose and out[t] is None:
            print(f"   no mask file for task {t}")
    return out


def _read_packed_tall_masks(mask_path: str) -> Dict[str, "torch.Tensor"]:
    """{task: uint8 packed stream} from an .npz (numpy.load) or a torch file of uint8 tensors -- loaders that
    execute nothing from the file."""
    import numpy as np
    packed = None
    try:
        z = np.load(mask_path, allow_pickle=False)
        if hasattr(z, "files"):
            packed = {k: z[k] for k in z.files}
    except Exception:
        packed = None
    if packed is None:
        try:
            packed = torch.load(mask_path, map_location="cpu", weights_only=True)
        except Exception as e:
            raise RuntimeError(f"{mask_path}: not loadable without unpickling arbitrary objects ({e}); "
                               "convert it with numpy.savez(path, **packed_masks)") from e
    out = {}
    for task, pm in packed.items():
        t = pm if isinstance(pm, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pm))
        if t.dtype != torch.uint8:
            raise TypeError(f"Unexpected type for packed_mask: {t.dtype}")
        out[task] = t.reshape(-1)
    return out


def combine_tall_masks_packed(mask_path: str, task_names: List[str], reference_state_dict: Dict[str, torch.Tensor],
                              strategy: str = "union", device="cuda", remove_keys: Optional[List[str]] = None
                              ) -> Dict[str, torch.Tensor]:
    """load_tall_mask_file + combine_masks (mask_loader.py:125-206, 488-648) without ever unpacking on the host:
    the bit-packed per-task streams go to the GPU as they are and ``svdq_maskset_combine_packed_indices`` votes on
    them in place (parameters = sorted keys of the reference state dict, each at its bit offset in the stream).
    Tasks missing from the file are skipped, as tasks with ``None`` masks are in combine_masks.
    Returns {parameter: combined bool mask on the GPU, shaped like the parameter}."""
    if strategy not in nat.MASK_STRATEGIES:
        raise ValueError(f"Unknown mask strategy: {strategy}")
    dev = resolve_device(device)
    packed = _read_packed_tall_masks(mask_path)
    streams = [packed[t] for t in task_names if t in packed]
    if not streams:
        return {}
    skip = set(remove_keys or [])
    keys = [k for k in sorted(reference_state_dict.keys()) if k not in skip]
    sizes = [reference_state_dict[k].numel() for k in keys]
    offs, acc = [], 0
    for n in sizes:
        offs.append(acc)
        acc += n
    live = [i for i, n in enumerate(sizes) if n > 0]
    out: Dict[str, torch.Tensor] = {k: torch.zeros(reference_state_dict[k].shape, dtype=torch.bool, device=dev)
                                    for k, n in zip(keys, sizes) if n == 0}
    if live:
        with torch.cuda.device(dev):
            ms = MaskSet([sizes[i] for i in live], dev)
            outs, _, _, _, _ = ms.prepare_combine_packed_indices(streams, [offs[i] for i in live], strategy,
                                                                 want_false=False)
            ms.run_combine_packed_indices()
            torch.cuda.current_stream().synchronize()
            ms.close()
        for i, o in zip(live, outs):
            out[keys[i]] = o.view(torch.bool).view(reference_state_dict[keys[i]].shape)
    return {k: out[k] for k in keys}
