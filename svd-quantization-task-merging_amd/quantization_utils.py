"""
The reference's top-level ``quantization_utils`` module on the GPU (reference quantization_utils.py:60-172):
same function names, arguments and returned tuples.  Single-tensor calls of the batched kernels in
csrc/svdq_ingest.hip (whole state dicts go through svdq_amd.ingest.quantize_state_dict in three launches).

``dequantize_absmax`` multiplies by the scale, as the reference does (quantization_utils.py:102-134).
``asymmetric_quantization`` with ``qbit = 16`` is not implemented (the reference stores codes up to 65535 in
int16 there).
"""
from __future__ import annotations

from typing import Tuple

import torch

from .ingest import dequantize_payloads, quantize_state_dict

FLOAT32_BITS = 32


def absmax_quantization(X: torch.Tensor, qbit: int = 8, verbose: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """quantization_utils.py:60-73: (int8 | int16 codes shaped like X, 0-d scale)."""
    p = quantize_state_dict({"x": X}, qbit, "absmax", X.device if X.is_cuda else "cuda",
                            skip_int64=False, skip_uint8=False)["x"]
    return p["quantized"], p["scale"]


def asymmetric_quantization(X: torch.Tensor, qbit: int = 8, verbose: bool = False
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """quantization_utils.py:76-99: (uint8 codes shaped like X, 0-d scale, 0-d zero_point)."""
    p = quantize_state_dict({"x": X}, qbit, "asymmetric", X.device if X.is_cuda else "cuda",
                            skip_int64=False, skip_uint8=False)["x"]
    return p["quantized"], p["scale"], p["zero_point"]


def dequantize_absmax(X_q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """quantization_utils.py:102-134: X_q.float() * scale."""
    return dequantize_payloads({"x": {"quantized": X_q, "scale": scale}}, "absmax",
                               X_q.device if X_q.is_cuda else "cuda", reshape=False)["x"]


def dequantize_asymmetric(X_q: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor) -> torch.Tensor:
    """quantization_utils.py:137-172: (X_q.float() - zero_point) / scale."""
    return dequantize_payloads({"x": {"quantized": X_q, "scale": scale, "zero_point": zero_point}}, "asymmetric",
                               X_q.device if X_q.is_cuda else "cuda", reshape=False)["x"]
