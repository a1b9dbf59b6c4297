"""
Coefficient-space merge on the GPU with the reference's callables (SURVEY.md section 8 f1;
reference src/svd_hybrid/merge.py:61-552).  The heavy step -- U c + mean over every row of a
parameter -- is `svdq_reconstruct` (HBM-bound: reads the fp16 basis once, writes fp32); the masked
scatter is `svdq_mask_expand`.  Averaging N x N scalars per parameter is host-sized work on small
device tensors.  Weights come from weighting.py, cluster labels from clustering.py (Gram-based).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _native as nat
from .pipeline import prepare_vector, resolve_device, wants_cpu, _ptr, _stream_ptr
from .rtvq import RTVQQuantizer


def dequantize_and_average(compressed_coeffs: Dict[str, Dict], weights: Dict[str, float], quantizer: RTVQQuantizer,
                           region: str = "masked", device: str = "cpu"
                           ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Reference merge.py:61-141: tasks sorted by name, weights renormalised over present tasks."""
    names = sorted(compressed_coeffs.keys())
    highs, lows, ws = [], [], []
    for name in names:
        art = compressed_coeffs[name]
        if art is None or art.get(region) is None:
            continue
        ra = art[region]
        highs.append(ra["c_high_fp16"].to(device).float())
        lows.append(quantizer.dequantize(ra["c_low_quant"], device=device).float())
        ws.append(weights.get(name, 1.0 / len(names)))
    if not highs:
        return None, None
    tot = sum(ws)
    w = torch.tensor([x / tot for x in ws], device=device, dtype=torch.float32).view(-1, 1)
    return (torch.stack(highs, dim=0) * w).sum(dim=0), (torch.stack(lows, dim=0) * w).sum(dim=0)


def reconstruct_from_coefficients(avg_c_high: torch.Tensor, avg_c_low: torch.Tensor, U_high: torch.Tensor,
                                  U_low: torch.Tensor, device: str = "cpu", mean: Optional[torch.Tensor] = None
                                  ) -> torch.Tensor:
    """Reference merge.py:144-194: U_high c_high + U_low c_low (+ mean); result on ``device`` (computed on the GPU)."""
    out = _reconstruct(avg_c_high, avg_c_low, U_high, U_low, mean, 1.0)
    return out.cpu() if wants_cpu(device) else out


def _reconstruct(avg_c_high, avg_c_low, U_high, U_low, mean, scale: float) -> torch.Tensor:
    """``scale`` folds the ``* noise_shrink`` of merge.py:270 into the same streaming pass."""
    lib = nat.lib()
    dev = resolve_device(U_high.device if U_high.is_cuda else "cuda")
    D = U_high.shape[0] if U_high.dim() == 2 else U_low.shape[0]
    k = U_high.shape[1] if U_high.dim() == 2 else 0
    nl = U_low.shape[1] if U_low.dim() == 2 else 0
    fp16 = (U_high.dtype == torch.float16) if k else (U_low.dtype == torch.float16)
    dt = torch.float16 if fp16 else torch.float32
    uh = U_high.to(device=dev, dtype=dt).contiguous() if k else None
    ul = U_low.to(device=dev, dtype=dt).contiguous() if nl else None
    coef = torch.cat([avg_c_high.to(dev).float().reshape(-1), avg_c_low.to(dev).float().reshape(-1)]).contiguous()
    if coef.numel() != k + nl:
        raise ValueError(f"Shape mismatch: {coef.numel()} coefficients for {k + nl} basis columns")
    m = prepare_vector(mean.squeeze() if mean.dim() > 1 else mean, dev) if mean is not None else None
    out = torch.empty(D, dtype=torch.float32, device=dev)
    if D == 0:
        return out
    with torch.cuda.device(dev):
        if k + nl <= 32:
            nat.check(lib.svdq_reconstruct(_ptr(uh), _ptr(ul), int(fp16), D, k, nl, _ptr(coef), _ptr(m), float(scale),
                                           _ptr(out), _stream_ptr()), "svdq_reconstruct")
        else:
            # wider than the path ever is (N <= 32): column groups, each adding onto the running sum through the
            # kernel's `mean` input; the scale goes on last
            groups = [(u[:, c0:c0 + 32].contiguous(), off + c0, min(32, n - c0))
                      for u, n, off in ((uh, k, 0), (ul, nl, k)) if n for c0 in range(0, n, 32)]
            acc, dst = m, (out, torch.empty_like(out))     # input and output of a launch never alias
            for i, (u, c0, cn) in enumerate(groups):
                last = i == len(groups) - 1
                nat.check(lib.svdq_reconstruct(_ptr(u), _ptr(None), int(fp16), D, cn, 0, _ptr(coef[c0:c0 + cn]),
                                               _ptr(acc), float(scale) if last else 1.0, _ptr(dst[i % 2]),
                                               _stream_ptr()), "svdq_reconstruct")
                acc = dst[i % 2]
            out = acc
    return out


def merge_parameter(param_name: str, compressed_params: Dict[str, Dict], basis: Dict, weights: Dict[str, float],
                    quantizer: RTVQQuantizer, original_shape: torch.Size, mask: Optional[torch.Tensor] = None,
                    include_noise: bool = False, noise_shrink: float = 1.0, device: str = "cpu") -> torch.Tensor:
    """Reference merge.py:197-301."""
    from .mask_loader import reconstruct_from_masked
    dev = resolve_device(device)
    merged_masked = merged_unmasked = None
    bm = basis.get("masked")
    if bm is not None:
        ch, cl = dequantize_and_average(compressed_params, weights, quantizer, region="masked", device=dev)
        if ch is not None:
            merged_masked = reconstruct_from_coefficients(ch, cl, bm["U_high"], bm["U_low"], device=dev,
                                                          mean=bm.get("mean"))
    if include_noise:
        bu = basis.get("noise")
        if bu is not None:
            ch, cl = dequantize_and_average(compressed_params, weights, quantizer, region="unmasked", device=dev)
            if ch is not None:
                merged_unmasked = _reconstruct(ch, cl, bu["U_high"], bu["U_low"], bu.get("mean"), noise_shrink)
    if mask is not None and merged_masked is not None:
        res = reconstruct_from_masked(merged_masked, merged_unmasked, mask, original_shape)
    elif merged_masked is not None:
        res = merged_masked.view(original_shape)
    else:
        res = torch.zeros(original_shape, device=dev)
    return res.cpu() if wants_cpu(device) else res      # the reference returns on `device` (default "cpu")


def merge_all_parameters(compressed_all: Dict[str, Dict[str, Dict]], bases: Dict[str, Dict],
                         masks: Dict[str, torch.Tensor], weights: Dict[str, float],
                         original_shapes: Dict[str, torch.Size], config, device: str = "cpu",
                         verbose: bool = True) -> Dict[str, torch.Tensor]:
    """Reference merge.py:304-426: parameters in sorted order, one quantizer for the run."""
    quantizer = RTVQQuantizer(num_bits=config.svd_low_bits, num_stages=config.svd_rtvq_stages)
    merged = {}
    for name in sorted(compressed_all.keys()):
        merged[name] = merge_parameter(name, compressed_all[name], bases[name], weights, quantizer,
                                       original_shapes[name], mask=masks.get(name),
                                       include_noise=config.svd_include_noise, noise_shrink=config.svd_noise_shrink,
                                       device=device)
    if verbose:
        print(f"   merged {len(merged)} parameters")
    return merged


def apply_merged_deltas(base_state_dict: Dict[str, torch.Tensor], merged_deltas: Dict[str, torch.Tensor],
                        device: str = "cpu", verbose: bool = True) -> Dict[str, torch.Tensor]:
    """Reference merge.py:429-552: merged[param] = base[param] + delta[param]; others are cloned."""
    out = {}
    for name, base in base_state_dict.items():
        if name in merged_deltas:
            out[name] = base + merged_deltas[name].to(base.device)
        else:
            out[name] = base.clone()
    if verbose:
        print(f"   applied {sum(n in merged_deltas for n in base_state_dict)} merged deltas")
    return out


def merge_with_clustering(compressed_all: Dict[str, Dict[str, Dict]], bases: Dict[str, Dict],
                          masks: Dict[str, torch.Tensor], weights: Dict[str, float],
                          cluster_assignments: Dict[str, int], original_shapes: Dict[str, torch.Size], config,
                          device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Reference merge.py:555-626: merge inside each cluster (weights renormalised over its members), then
    average the cluster results with softmax(mean member weight) shares (clustering.py:374-425)."""
    from .clustering import get_cluster_members, merge_cluster_results
    clusters = get_cluster_members(cluster_assignments)
    per_cluster, performance = {}, {}
    for cid, members in clusters.items():
        w = {m: weights.get(m, 1.0) for m in members}
        total = sum(w.values())
        w = {m: v / total for m, v in w.items()}
        subset = {p: {m: arts[m] for m in members if m in arts} for p, arts in compressed_all.items()}
        per_cluster[cid] = merge_all_parameters(subset, bases, masks, w, original_shapes, config, device,
                                                verbose=False)
        performance[cid] = sum(weights.get(m, 1.0) for m in members) / len(members)
    return merge_cluster_results(per_cluster, performance, device)
