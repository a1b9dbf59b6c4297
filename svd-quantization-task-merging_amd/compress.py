"""
Task-vector compression with the reference's call signatures and return layouts
(reference src/svd_hybrid/compress.py:6-207).

Two routes produce identical artifact dicts:
  * bases that came out of ``driver.run_basis_and_compress`` / ``driver.build_bases`` carry the
    coefficients the fused pass 2 already computed; ``compress_all_parameters`` just assembles them;
  * any other basis (a caller's own U) goes through ``svdq_project`` + the GPU quantizer, one
    (parameter, task) at a time, exactly like the reference's loops.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _native as nat
from .pipeline import prepare_vector, resolve_device, _ptr, _stream_ptr
from .rtvq import RTVQQuantizer


def project_to_basis(delta: torch.Tensor, U_high: torch.Tensor, U_low: torch.Tensor
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference compress.py:6-21: (U_high.float().T @ delta, U_low.float().T @ delta)."""
    return _project(delta, U_high, U_low, None)


def _project(delta, U_high, U_low, mean):
    lib = nat.lib()
    dev = resolve_device(U_high.device if U_high.is_cuda else (delta.device if delta.is_cuda else "cuda"))
    out_dev = delta.device
    x = prepare_vector(delta, dev)
    D = x.numel()
    k = U_high.shape[1] if U_high.dim() == 2 else 0
    nl = U_low.shape[1] if U_low.dim() == 2 else 0
    if k + nl == 0:
        return torch.zeros(0, device=out_dev), torch.zeros(0, device=out_dev)
    if (k and U_high.shape[0] != D) or (nl and U_low.shape[0] != D):
        raise ValueError(f"Shape mismatch: basis rows vs delta length {D}")
    fp16 = (U_high.dtype == torch.float16) if k else (U_low.dtype == torch.float16)
    dt = torch.float16 if fp16 else torch.float32
    uh = U_high.to(device=dev, dtype=dt).contiguous() if k else None
    ul = U_low.to(device=dev, dtype=dt).contiguous() if nl else None
    m = prepare_vector(mean.squeeze() if mean.dim() > 1 else mean, dev) if mean is not None else None
    if k + nl <= 32:
        c = _project_cols(lib, dev, uh, ul, fp16, D, k, nl, x, m)
    else:
        # the kernel takes up to 32 columns (the path never has more than N <= 32); a caller's own wider basis
        # (the reference's tests project onto full square bases) goes through in column groups
        parts = []
        for u, n in ((uh, k), (ul, nl)):
            for c0 in range(0, n, 32):
                cn = min(32, n - c0)
                parts.append(_project_cols(lib, dev, u[:, c0:c0 + cn].contiguous(), None, fp16, D, cn, 0, x, m))
        c = torch.cat(parts)
    c = c.to(out_dev)
    return c[:k], c[k:]


def _project_cols(lib, dev, uh, ul, fp16, D, k, nl, x, m):
    c = torch.empty(k + nl, dtype=torch.float32, device=dev)
    work = torch.empty(int(lib.svdq_project_work_bytes(D, k + nl)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_project(_ptr(uh), _ptr(ul), int(fp16), D, k, nl, _ptr(x), _ptr(m), _ptr(c), _ptr(work),
                                   _stream_ptr()), "svdq_project")
    return c


def compress_single_task(task_delta: torch.Tensor, U_high: torch.Tensor, U_low: torch.Tensor,
                         quantizer: RTVQQuantizer, device: str = "cpu", mean: Optional[torch.Tensor] = None) -> Dict:
    """Reference compress.py:24-56."""
    c_high, c_low = _project(task_delta, U_high, U_low, mean)
    c_high_fp16 = c_high.half() if c_high.dtype != torch.float16 else c_high
    return {"c_high_fp16": c_high_fp16.cpu(), "c_low_quant": quantizer.quantize(c_low.cpu())}


def compress_masked_regions(task_deltas_masked: Dict[str, torch.Tensor],
                            task_deltas_unmasked: Optional[Dict[str, torch.Tensor]], basis_masked: Dict,
                            basis_unmasked: Optional[Dict], quantizer: RTVQQuantizer, device: str = "cpu"
                            ) -> Dict[str, Dict]:
    """Reference compress.py:59-111."""
    out = {}
    mean_m = basis_masked.get("mean") if basis_masked is not None else None
    mean_u = basis_unmasked.get("mean") if basis_unmasked is not None else None
    for task, dm in task_deltas_masked.items():
        art = {}
        if basis_masked is not None and len(dm) > 0:
            art["masked"] = compress_single_task(dm, basis_masked["U_high"], basis_masked["U_low"], quantizer, device,
                                                 mean=mean_m)
        else:
            art["masked"] = None
        if (basis_unmasked is not None and task_deltas_unmasked is not None and task in task_deltas_unmasked
                and len(task_deltas_unmasked[task]) > 0):
            art["unmasked"] = compress_single_task(task_deltas_unmasked[task], basis_unmasked["U_high"],
                                                   basis_unmasked["U_low"], quantizer, device, mean=mean_u)
        else:
            art["unmasked"] = None
        out[task] = art
    return out


def compress_parameter(param_name: str, task_vectors: Dict[str, Dict[str, torch.Tensor]],
                       mask: Optional[torch.Tensor], basis: Dict, quantizer: RTVQQuantizer,
                       include_noise: bool = False, min_mask_size: int = 10, device: str = "cpu") -> Optional[Dict]:
    """Reference compress.py:114-170."""
    from .mask_loader import apply_mask_to_tensor, get_unmasked_portion
    deltas = {t: tv[param_name] for t, tv in task_vectors.items() if param_name in tv}
    if not deltas:
        return None
    masked, unmasked = {}, {}
    for t, d in deltas.items():
        if mask is not None and mask.shape == d.shape:
            if mask.sum() >= min_mask_size:
                masked[t] = apply_mask_to_tensor(d, mask)
            else:
                masked[t] = torch.tensor([])
            if include_noise:
                unmasked[t] = get_unmasked_portion(d, mask)
        else:
            masked[t] = d.flatten()
    return compress_masked_regions(masked, unmasked if include_noise else None, basis.get("masked"),
                                   basis.get("noise") if include_noise else None, quantizer, device)


def compress_all_parameters(task_vectors: Dict[str, Dict[str, torch.Tensor]], masks: Dict[str, torch.Tensor],
                            bases: Dict[str, Dict], config, device: str = "cpu") -> Dict[str, Dict]:
    """Reference compress.py:173-207.  Parameters in sorted order, tasks in insertion order, one
    quantizer (b, S) for the whole run."""
    import gc
    from .driver import artifacts_from_batch
    quantizer = RTVQQuantizer(num_bits=config.svd_low_bits, num_stages=config.svd_rtvq_stages)
    out = {}
    # assembling ~10^4 small dictionaries and tensors: keep the cyclic collector from re-walking them all
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        for name in sorted(bases.keys()):
            basis = bases[name]
            fused = artifacts_from_batch(name, basis, task_vectors, config)
            if fused is not None:
                out[name] = fused
                continue
            comp = compress_parameter(name, task_vectors, masks.get(name), basis, quantizer,
                                      include_noise=config.svd_include_noise,
                                      min_mask_size=config.svd_min_mask_size, device=device)
            if comp is not None:
                out[name] = comp
    finally:
        if gc_was_on:
            gc.enable()
    return out
