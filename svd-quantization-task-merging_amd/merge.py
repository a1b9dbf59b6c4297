"""
Parity consumers of the artifacts (reference src/svd_hybrid/merge.py:61-194).
They define the reconstruction that "recon MSE vs ref" is measured on.  Not on the timed path:
device tensor ops, same arithmetic order as the reference.  The full merge (weights, clusters,
apply_merged_deltas) is a later row of SURVEY.md section 8(f).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

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
    """Reference merge.py:144-194: U_high c_high + U_low c_low (+ mean)."""
    Uh = U_high.to(device).float()
    Ul = U_low.to(device).float()
    out = Uh @ avg_c_high.to(device) + Ul @ avg_c_low.to(device)
    if mean is not None:
        out = out + mean.squeeze().to(device).float()
    return out
