"""
The reference's top-level ``quantization_utils`` module on the GPU (reference quantization_utils.py:60-172):
same function names, arguments and returned tuples.  Single-tensor calls of the batched kernels in
csrc/svdq_ingest.hip (whole state dicts go through svdq_amd.ingest.quantize_state_dict in three launches).

``dequantize_absmax`` multiplies by the scale, as the reference does (quantization_utils.py:102-134).
``asymmetric_quantization`` with ``qbit = 16`` gives int16 codes exactly as the reference's cast leaves them (values
above 32767 wrap negative; svdq_asym16_quantize).  The two print-only error checkers of the reference
(quantization_utils.py:175-212) are here under their original names, typo included.
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
    """quantization_utils.py:76-99: (uint8 codes shaped like X, 0-d scale, 0-d zero_point); int16 codes for qbit = 16."""
    if qbit == 16:
        from .rtvq import asymmetric_quantization as _asym
        return _asym(X, 16)
    p = quantize_state_dict({"x": X}, qbit, "asymmetric", X.device if X.is_cuda else "cuda",
                            skip_int64=False, skip_uint8=False)["x"]
    return p["quantized"], p["scale"], p["zero_point"]


def dequantize_absmax(X_q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """quantization_utils.py:102-134: X_q.float() * scale."""
    return dequantize_payloads({"x": {"quantized": X_q, "scale": scale}}, "absmax",
                               X_q.device if X_q.is_cuda else "cuda", reshape=False)["x"]


def dequantize_asymmetric(X_q: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor) -> torch.Tensor:
    """quantization_utils.py:137-172: (X_q.float() - zero_point) / scale."""
    if X_q.dtype == torch.int16:
        from .rtvq import asymmetric_dequantization as _deq
        return _deq(X_q, scale, zero_point)
    return dequantize_payloads({"x": {"quantized": X_q, "scale": scale, "zero_point": zero_point}}, "asymmetric",
                               X_q.device if X_q.is_cuda else "cuda", reshape=False)["x"]


def _accumulated_error(original_state_dict, quantized_state_dict, code_dtype, with_zero_point: bool):
    """Sum over all entries of |original - reconstructed|, as the reference's two checkers compute it: entries of
    ``code_dtype`` are rebuilt from ``<key>_qscale`` (and ``<key>_qzeropoint``), everything else is compared as it is.
    Note the reference DIVIDES int8 codes by the scale here (quantization_utils.py:185), unlike dequantize_absmax."""
    total = 0
    for key, w in original_state_dict.items():
        q = quantized_state_dict[key]
        if q.dtype == code_dtype and key + "_qscale" in quantized_state_dict:
            scale = quantized_state_dict[key + "_qscale"]
            if with_zero_point:
                rec = (q.to(torch.float) - quantized_state_dict[key + "_qzeropoint"].to(torch.float)) / scale
            else:
                rec = q.to(torch.float) / scale
        else:
            rec = q
        total = total + torch.sum(torch.abs(w - rec.to(w.device)))
    return total


def qunatization_error_check(original_state_dict, quantized_state_dict):
    """quantization_utils.py:175-192 (name as in the reference): prints the accumulated absolute error of an
    absmax-quantized state dict (int8 entries + ``<key>_qscale``)."""
    print(f"accumuated Quantized error: {_accumulated_error(original_state_dict, quantized_state_dict, torch.int8, False)}")


def quantization_error_check_asymmetric(original_state_dict, quantized_state_dict):
    """quantization_utils.py:195-212: the same for asymmetric entries (uint8 + ``_qscale`` + ``_qzeropoint``)."""
    print(f"accumuated Quantized error: {_accumulated_error(original_state_dict, quantized_state_dict, torch.uint8, True)}")
