"""
Multi-stage residual affine quantizer ("RTVQ") on the GPU -- same names, arguments, defaults
and return layouts as the reference's src/svd_hybrid/rtvq.py (and the asymmetric functions of
quantization_utils.py:76-99, 137-172), computed by libsvdq_hip.so.

Payload tensors are CPU tensors, as in the reference (rtvq.py:71-73).  Degenerate inputs
(one element, all elements equal) give scale = inf and NaN downstream exactly like the
reference: this package is bug-compatible there on purpose.
"""
from __future__ import annotations

from ctypes import c_void_p
from typing import Dict, List, Tuple

import torch

from . import _native as nat
from .pipeline import prepare_vector, resolve_device, _ptr, _stream_ptr


def _quantize_device(x: torch.Tensor, bits: int, stages: int):
    """x: flat fp32 device tensor, n >= 1 -> (codes [stages, n] u8, scale, zp, rnorm [stages]) on device."""
    lib = nat.lib()
    n = x.numel()
    dev = x.device
    stride = (n + 3) // 4 * 4
    codes = torch.empty((stages, stride), dtype=torch.uint8, device=dev)
    scale = torch.empty(stages, dtype=torch.float32, device=dev)
    zp = torch.empty(stages, dtype=torch.float32, device=dev)
    rnorm = torch.empty(stages, dtype=torch.float32, device=dev)
    work = torch.empty(int(lib.svdq_rtvq_work_bytes(n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_rtvq_quantize(_ptr(x), n, bits, stages, _ptr(codes), stride, _ptr(scale), _ptr(zp),
                                         _ptr(rnorm), _ptr(work), _stream_ptr()), "svdq_rtvq_quantize")
    return codes[:, :n], scale, zp, rnorm


def asymmetric_quantization(X: torch.Tensor, qbit: int = 8, verbose: bool = False
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Reference rtvq.py:4-27: (uint8 codes shaped like X, 0-d scale, 0-d zero_point), on X's device."""
    if qbit > 8 and qbit != 16:
        # the reference falls off the end of its dtype selection here (UnboundLocalError, rtvq.py:22-27)
        raise ValueError(f"qbit must be <= 8 or 16, got {qbit}")
    dev = resolve_device(X.device if X.is_cuda else "cuda")
    x = prepare_vector(X, dev)
    if qbit == 16:   # int16 codes, exactly as the reference's cast leaves them (values above 32767 wrap negative)
        lib = nat.lib()
        n = x.numel()
        codes = torch.empty(n, dtype=torch.int16, device=dev)
        sz = torch.empty(2, dtype=torch.float32, device=dev)
        work = torch.empty(int(lib.svdq_rtvq_work_bytes(n)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.svdq_asym16_quantize(_ptr(x), n, _ptr(codes), _ptr(sz[0:1]), _ptr(sz[1:2]), _ptr(work),
                                               _stream_ptr()), "svdq_asym16_quantize")
        return codes.view(X.shape).to(X.device), sz[0].to(X.device), sz[1].to(X.device)
    codes, scale, zp, _ = _quantize_device(x, qbit, 1)
    out_dev = X.device
    return codes[0].contiguous().view(X.shape).to(out_dev), scale[0].to(out_dev), zp[0].to(out_dev)


def asymmetric_dequantization(quantized: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor) -> torch.Tensor:
    """Reference rtvq.py:29-36: (q.float() - zero_point) / scale."""
    if quantized.dtype == torch.int16:
        lib = nat.lib()
        out_dev = quantized.device
        dev = resolve_device(out_dev if quantized.is_cuda else "cuda")
        q = quantized.to(dev).contiguous().view(-1)
        sz = torch.stack([torch.as_tensor(scale, dtype=torch.float32).reshape(()),
                          torch.as_tensor(zero_point, dtype=torch.float32).reshape(())]).to(dev)
        out = torch.empty(q.numel(), dtype=torch.float32, device=dev)
        if q.numel():
            with torch.cuda.device(dev):
                nat.check(lib.svdq_asym16_dequantize(_ptr(q), q.numel(), _ptr(sz[0:1]), _ptr(sz[1:2]), _ptr(out),
                                                     _stream_ptr()), "svdq_asym16_dequantize")
        return out.view(quantized.shape).to(out_dev)
    return _dequantize([quantized], [scale], [zero_point], quantized.device)


def _dequantize(codes_list: List[torch.Tensor], scales, zps, device) -> torch.Tensor:
    lib = nat.lib()
    out_dev = torch.device(device) if not isinstance(device, torch.device) else device
    dev = resolve_device(out_dev)
    shape = codes_list[0].shape
    n = codes_list[0].numel()
    stages = len(codes_list)
    stride = (n + 3) // 4 * 4
    codes = torch.zeros((stages, stride), dtype=torch.uint8, device=dev)
    for s, c in enumerate(codes_list):
        codes[s, :n] = c.to(dev).reshape(-1)
    sc = torch.stack([torch.as_tensor(s, dtype=torch.float32).reshape(()) for s in scales]).to(dev)
    zp = torch.stack([torch.as_tensor(z, dtype=torch.float32).reshape(()) for z in zps]).to(dev)
    out = torch.empty(stride, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_rtvq_dequantize(_ptr(codes), stride, n, stages, _ptr(sc), _ptr(zp), _ptr(out),
                                           _stream_ptr()), "svdq_rtvq_dequantize")
    return out[:n].view(shape).to(out_dev)


def multistage_residual_quantization(tensor: torch.Tensor, num_bits: int = 4, num_stages: int = 2,
                                     verbose: bool = False) -> List[Dict]:
    """Reference rtvq.py:39-82. Empty tensor -> []."""
    if tensor.numel() == 0:
        return []
    dev = resolve_device(tensor.device if tensor.is_cuda else "cuda")
    x = prepare_vector(tensor, dev)
    codes, scale, zp, rnorm = _quantize_device(x, num_bits, num_stages)
    codes_h, scale_h, zp_h, rnorm_h = codes.cpu(), scale.cpu(), zp.cpu(), rnorm.cpu()
    return [{"stage": s,
             "quantized": codes_h[s].contiguous().view(tensor.shape),
             "scale": scale_h[s].clone(),
             "zero_point": zp_h[s].clone(),
             "residual_norm": float(rnorm_h[s])} for s in range(num_stages)]


def multistage_residual_dequantization(payloads: List[Dict], device: str = "cpu") -> torch.Tensor:
    """Reference rtvq.py:85-103. Empty payload list -> torch.tensor([])."""
    if not payloads:
        return torch.tensor([], device=device)
    return _dequantize([p["quantized"] for p in payloads], [p["scale"] for p in payloads],
                       [p["zero_point"] for p in payloads], device)


class RTVQQuantizer:
    """Reference rtvq.py:106-139."""

    def __init__(self, num_bits: int = 4, num_stages: int = 2):
        self.num_bits = num_bits
        self.num_stages = num_stages

    def quantize(self, tensor: torch.Tensor) -> Dict:
        payloads = multistage_residual_quantization(tensor, num_bits=self.num_bits, num_stages=self.num_stages)
        return {"payloads": payloads, "num_bits": self.num_bits, "num_stages": self.num_stages,
                "original_shape": tensor.shape, "original_dtype": str(tensor.dtype)}

    def dequantize(self, quantized_obj: Dict, device: str = "cpu") -> torch.Tensor:
        result = multistage_residual_dequantization(quantized_obj["payloads"], device=device)
        if "original_shape" in quantized_obj:
            result = result.view(quantized_obj["original_shape"])
        return result


def estimate_compression_ratio(original: torch.Tensor, quantized_obj: Dict) -> float:
    """Reference rtvq.py:142-161 (host arithmetic only)."""
    original_size = original.numel() * 4
    stages, bits = quantized_obj["num_stages"], quantized_obj["num_bits"]
    compressed = original.numel() * bits / 8 * stages + 8 * stages
    return original_size / max(compressed, 1)
