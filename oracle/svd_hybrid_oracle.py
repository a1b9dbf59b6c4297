"""
oracle/svd_hybrid_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's per-parameter SVD-Hybrid compressor hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does (it raises if its HIP
library is missing instead of falling back here).

What is restated (paths relative to /root/reference):

  R1  mask combine            src/svd_hybrid/mask_loader.py:412-485, 488-648
  R2  mask apply / complement src/svd_hybrid/mask_loader.py:651-709
  R3  stack + centre          src/svd_hybrid/basis.py:63-113
  R4  thin SVD                src/svd_hybrid/basis.py:216-249  (torch.linalg.svd -> LAPACK gesdd)
  R5  energy / rank           src/svd_hybrid/basis.py:116-213
  R6  basis split             src/svd_hybrid/basis.py:252-409, 412-468
  R7  fp16 cast of the basis  src/svd_hybrid/cli.py:354-361      (before projection)
  R8  projection              src/svd_hybrid/compress.py:6-21
  R9  per-task compress       src/svd_hybrid/compress.py:24-56
  R10-R12 quantizer           src/svd_hybrid/rtvq.py:4-139        (C restatement: rtvq_oracle.c)
  R13 loops / dict layout     src/svd_hybrid/compress.py:59-207
  R14 reconstruction          src/svd_hybrid/merge.py:61-194, mask_loader.py:712-763
  f2  per-task error tuple    src/svd_hybrid/diagnostics.py:72-117, 186-215 (no mean added back)

Third-party arithmetic: the SVD itself is LAPACK ``gesdd`` reached through
``torch.linalg.svd`` (MKL in this image's torch 2.10.0 CPU build); the reference pins
no torch/MKL version.  The oracle calls the same routine, so on one machine it agrees
with the reference to the last bit for everything except BLAS reduction order.

Pinned: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which ``tests/golden/make_golden.py`` produced by importing the
reference itself in the build container, and against the reference's own KATs
(tests/test_rank_selection.py, test_mask_strategies.py, test_rtvq.py, test_mean_handling.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librtvq_oracle.so")
_lib = None


def build_c_oracle(force: bool = False) -> str:
    """Compile rtvq_oracle.c with gcc (oracle/Makefile). Returns the .so path."""
    src = os.path.join(_HERE, "rtvq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _clib():
    global _lib
    if _lib is None:
        build_c_oracle()
        lib = ctypes.CDLL(_LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        lib.oracle_asym_quantize.argtypes = [f32p, ctypes.c_size_t, ctypes.c_int, u8p, f32p, f32p]
        lib.oracle_asym_dequantize.argtypes = [u8p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, f32p]
        lib.oracle_rtvq_quantize.argtypes = [f32p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                             u8p, f32p, f32p, f32p, f32p]
        lib.oracle_rtvq_dequantize.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, f32p, f32p, f32p]
        for fn in (lib.oracle_asym_quantize, lib.oracle_asym_dequantize,
                   lib.oracle_rtvq_quantize, lib.oracle_rtvq_dequantize):
            fn.restype = None
        _lib = lib
    return _lib


def _f32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


# --------------------------------------------------------------------------- R10-R12
def asym_quantize(x, bits: int) -> Tuple[np.ndarray, np.float32, np.float32]:
    """rtvq.py:4-27 (== quantization_utils.py:76-99). Returns (codes u8, scale, zero_point)."""
    xf = _f32(x).reshape(-1)
    q = np.empty(xf.size, dtype=np.uint8)
    sc = ctypes.c_float()
    zp = ctypes.c_float()
    _clib().oracle_asym_quantize(_ptr(xf, ctypes.c_float), xf.size, bits, _ptr(q, ctypes.c_uint8),
                                 ctypes.byref(sc), ctypes.byref(zp))
    return q.reshape(np.shape(x)), np.float32(sc.value), np.float32(zp.value)


def asym_dequantize(q, scale, zp) -> np.ndarray:
    """rtvq.py:29-36."""
    qq = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1)
    out = np.empty(qq.size, dtype=np.float32)
    _clib().oracle_asym_dequantize(_ptr(qq, ctypes.c_uint8), qq.size, float(scale), float(zp),
                                   _ptr(out, ctypes.c_float))
    return out.reshape(np.shape(q))


def absmax_quantize(x, bits: int = 8) -> Tuple[np.ndarray, np.float32]:
    """quantization_utils.py:60-73: s = (2^(b-1) - 1) / max|X| (python-int / Tensor = reciprocal * int, two
    roundings), codes = round(s * X) as int8 (int16 for 16 bits; half-to-even; no clamp)."""
    xf = _f32(x)
    amax = np.float32(np.max(np.abs(xf)))
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.float32(np.float32(1.0) / amax) * np.float32(2 ** (bits - 1) - 1)
        q = np.rint((s * xf).astype(np.float32))
    return q.astype(np.int16 if bits == 16 else np.int8), np.float32(s)


def absmax_dequantize(q, scale) -> np.ndarray:
    """quantization_utils.py:102-134: X_q.float() * scale (the reference multiplies; kept)."""
    return (np.asarray(q).astype(np.float32) * np.float32(scale)).astype(np.float32)


def task_vector(base: Dict[str, np.ndarray], finetuned: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """task_vector_loader.py:103-141: finetuned - base for keys in both with equal shapes (base's order)."""
    out = {}
    for k, b in base.items():
        if k in finetuned and np.shape(finetuned[k]) == np.shape(b):
            out[k] = np.asarray(finetuned[k]) - np.asarray(b)
    return out


def rtvq_quantize(x, bits: int = 4, stages: int = 2) -> Dict:
    """
    rtvq.py:39-82 + RTVQQuantizer.quantize (rtvq.py:111-126), array form:
    {"codes": u8 [stages, n], "scale": f32 [stages], "zero_point": f32 [stages],
     "residual_norm": f32 [stages], "shape": tuple, "num_bits", "num_stages"}.
    An empty input gives stages-less arrays (the reference returns payloads == []).
    """
    xf = _f32(x).reshape(-1)
    n = xf.size
    if n == 0:
        return {"codes": np.zeros((0, 0), np.uint8), "scale": np.zeros(0, np.float32),
                "zero_point": np.zeros(0, np.float32), "residual_norm": np.zeros(0, np.float32),
                "shape": tuple(np.shape(x)), "num_bits": bits, "num_stages": stages}
    codes = np.empty((stages, n), dtype=np.uint8)
    scale = np.empty(stages, dtype=np.float32)
    zp = np.empty(stages, dtype=np.float32)
    norm = np.empty(stages, dtype=np.float32)
    work = np.empty(n, dtype=np.float32)
    _clib().oracle_rtvq_quantize(_ptr(xf, ctypes.c_float), n, bits, stages,
                                 _ptr(codes, ctypes.c_uint8), _ptr(scale, ctypes.c_float),
                                 _ptr(zp, ctypes.c_float), _ptr(norm, ctypes.c_float),
                                 _ptr(work, ctypes.c_float))
    return {"codes": codes, "scale": scale, "zero_point": zp, "residual_norm": norm,
            "shape": tuple(np.shape(x)), "num_bits": bits, "num_stages": stages}


def rtvq_dequantize(obj: Dict) -> np.ndarray:
    """rtvq.py:85-103 + RTVQQuantizer.dequantize (rtvq.py:128-139)."""
    codes = np.ascontiguousarray(obj["codes"], dtype=np.uint8)
    if codes.size == 0:
        return np.zeros(0, dtype=np.float32)
    stages, n = codes.shape
    out = np.empty(n, dtype=np.float32)
    sc = _f32(obj["scale"])
    zp = _f32(obj["zero_point"])
    _clib().oracle_rtvq_dequantize(_ptr(codes, ctypes.c_uint8), n, stages,
                                   _ptr(sc, ctypes.c_float), _ptr(zp, ctypes.c_float),
                                   _ptr(out, ctypes.c_float))
    return out.reshape(obj.get("shape", (n,)))


def rtvq_quantize_numpy(x, bits: int = 4, stages: int = 2) -> Dict:
    """Second, numpy-only statement of the same arithmetic (cross-checks the C build flags)."""
    r = _f32(x).reshape(-1).copy()
    qmax = np.float32((1 << bits) - 1)
    out = {"codes": [], "scale": [], "zero_point": [], "residual_norm": []}
    with np.errstate(all="ignore"):
        for _ in range(stages):
            out["residual_norm"].append(np.float32(np.sqrt(np.sum(r.astype(np.float64) ** 2))))
            mn, mx = (np.float32(np.nan),) * 2 if np.isnan(r).any() else (r.min(), r.max())
            scale = np.float32(np.float32(np.float32(1.0) / np.float32(mx - mn)) * qmax)
            zp = np.float32(-1.0) * np.rint(np.float32(scale * mn))
            v = np.rint((scale * r).astype(np.float32) + zp)
            q = np.where(np.isnan(v), np.float32(0), np.clip(v, 0, qmax)).astype(np.uint8)
            deq = ((q.astype(np.float32) - zp).astype(np.float32) / scale).astype(np.float32)
            r = (r - deq).astype(np.float32)
            out["codes"].append(q)
            out["scale"].append(scale)
            out["zero_point"].append(np.float32(zp))
    return {"codes": np.stack(out["codes"]), "scale": np.array(out["scale"], np.float32),
            "zero_point": np.array(out["zero_point"], np.float32),
            "residual_norm": np.array(out["residual_norm"], np.float32)}


# --------------------------------------------------------------------------- R1, R2, R14 (masks)
def combine_masks(masks: Sequence[torch.Tensor], strategy: str) -> torch.Tensor:
    """mask_loader.py:412-485 / :604-611. masks: the per-task bool masks of ONE parameter."""
    if not masks:
        raise ValueError("Empty mask list")
    if strategy == "union":
        out = masks[0].clone()
        for m in masks[1:]:
            out = out | m
        return out
    if strategy == "intersection":
        out = masks[0].clone()
        for m in masks[1:]:
            out = out & m
        return out
    if strategy == "majority":
        votes = torch.stack([m.float() for m in masks], dim=0).sum(dim=0)
        return votes >= (0.5 * len(masks))
    raise ValueError(f"Unknown mask strategy: {strategy}")


def select_masked(t: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """mask_loader.py:651-679: order-preserving compaction of the True positions."""
    if t.shape != mask.shape:
        raise ValueError(f"Shape mismatch: tensor {t.shape} vs mask {mask.shape}")
    return t.flatten()[mask.flatten()]


def select_unmasked(t: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """mask_loader.py:682-709."""
    if t.shape != mask.shape:
        raise ValueError(f"Shape mismatch: tensor {t.shape} vs mask {mask.shape}")
    return t.flatten()[~mask.flatten()]


def scatter_masked(signal: torch.Tensor, noise: Optional[torch.Tensor], mask: torch.Tensor,
                   shape) -> torch.Tensor:
    """mask_loader.py:712-763."""
    fm = mask.flatten()
    out = torch.zeros(fm.numel(), dtype=signal.dtype)
    out[fm] = signal
    if noise is not None:
        out[~fm] = noise
    return out.view(shape)


# --------------------------------------------------------------------------- R3-R7 (basis)
def stack_center(vectors: Sequence[torch.Tensor], center: bool):
    """basis.py:63-113: T = stack(dim=1) [D,N]; mean over tasks [D,1]; T -= mean."""
    if not vectors:
        raise ValueError("Empty vector list")
    T = torch.stack(list(vectors), dim=1)
    mean = None
    if center:
        mean = T.mean(dim=1, keepdim=True)
        T = T - mean
    return T, mean


def energy_spectrum(S: torch.Tensor) -> torch.Tensor:
    """basis.py:116-156 (fp32 throughout)."""
    e = S ** 2
    tot = e.sum()
    if tot < 1e-10:
        return torch.ones_like(e)
    return torch.cumsum(e, dim=0) / tot


def select_rank(S: torch.Tensor, energy_threshold: float = 0.90, max_rank: Optional[int] = None,
                min_rank: int = 1) -> int:
    """basis.py:159-213: k = #(cum < thr) + 1, then min_rank / max_rank / len(S) clamps."""
    cum = energy_spectrum(S)
    k = int((cum < energy_threshold).sum().item()) + 1
    k = max(k, min_rank)
    if max_rank is not None:
        k = min(k, max_rank)
    return min(k, len(S))


def svd_basis(deltas: Sequence[torch.Tensor], energy_threshold: float = 0.90,
              max_rank: Optional[int] = None, center: bool = True, fp16: bool = False) -> Dict:
    """
    basis.py:252-409 (R3->R4->R5->R6) and, when ``fp16``, the cast of cli.py:354-361 (R7).
    Returns the reference's basis dict (tensors on CPU).
    """
    if not deltas:
        raise ValueError("Empty delta list")
    T, mean = stack_center([d.float() for d in deltas], center)
    D, N = T.shape
    U, S, _ = torch.linalg.svd(T, full_matrices=False)
    k = select_rank(S, energy_threshold, max_rank)
    U_high = U[:, :k].contiguous()
    U_low = U[:, k:].contiguous()
    energy_retained = energy_spectrum(S)[k - 1].item() if k > 0 else 0
    if fp16:
        U_high, U_low = U_high.half(), U_low.half()
    return {"U_high": U_high, "U_low": U_low, "singular_values": S, "k": k, "mean": mean,
            "energy_retained": energy_retained, "D": D, "N": N}


# --------------------------------------------------------------------------- R8, R9, R14
def project(delta: torch.Tensor, U_high: torch.Tensor, U_low: torch.Tensor):
    """compress.py:6-21: two fp32 GEMVs against the (possibly fp16-rounded) basis."""
    d = delta.float()
    return U_high.float().T @ d, U_low.float().T @ d


def compress_task(delta: torch.Tensor, U_high: torch.Tensor, U_low: torch.Tensor,
                  bits: int, stages: int, mean: Optional[torch.Tensor]) -> Dict:
    """compress.py:24-56: subtract mean, project, fp16(c_high), RTVQ(c_low)."""
    d = delta if mean is None else delta - mean.squeeze()
    c_high, c_low = project(d, U_high, U_low)
    return {"c_high": c_high, "c_low": c_low, "c_high_fp16": c_high.half(),
            "c_low_quant": rtvq_quantize(c_low.numpy(), bits, stages)}


def reconstruct(c_high: torch.Tensor, c_low: torch.Tensor, U_high: torch.Tensor,
                U_low: torch.Tensor, mean: Optional[torch.Tensor]) -> torch.Tensor:
    """merge.py:144-194: U_high c_high + U_low c_low (+ mean)."""
    out = U_high.float() @ c_high + U_low.float() @ c_low
    if mean is not None:
        out = out + mean.squeeze().float()
    return out


DIAG_KEYS = ("absolute_error", "relative_error", "max_absolute_error", "mean_absolute_error", "original_norm",
             "reconstructed_norm")


def reconstruction_error(original: torch.Tensor, reconstructed: torch.Tensor) -> Dict[str, float]:
    """diagnostics.py:72-117, in the dtype of the inputs (the reference runs it on fp32 tensors)."""
    error = original - reconstructed
    original_norm = original.norm().item()
    error_norm = error.norm().item()
    return {"absolute_error": error_norm,
            "relative_error": error_norm / original_norm if original_norm > 1e-10 else 0,
            "max_absolute_error": error.abs().max().item(),
            "mean_absolute_error": error.abs().mean().item(),
            "original_norm": original_norm,
            "reconstructed_norm": reconstructed.norm().item()}


def parameter_task_diagnostics(x: torch.Tensor, U_high: torch.Tensor, U_low: torch.Tensor, c_high: torch.Tensor,
                               c_low: torch.Tensor, dtype=torch.float32, mean: Optional[torch.Tensor] = None) -> Dict[str, float]:
    """diagnostics.py:186-215 for one (parameter, task): ``U_high.float() @ c_high + U_low.float() @ c_low`` -- the mean is
    NOT added back (``mean`` exists for the add_mean extension of the plan-level kernel) -- compared with the
    original (masked) delta.  ``dtype`` float32 is the reference's arithmetic; float64 evaluates the same formula on the
    same stored numbers without the fp32 rounding of the matrix products and sums (the checker of the HIP kernels, whose
    sums run in fp64)."""
    rec = U_high.to(dtype) @ c_high.to(dtype) + U_low.to(dtype) @ c_low.to(dtype)
    if mean is not None:
        rec = rec + mean.reshape(-1).to(dtype)
    return reconstruction_error(x.reshape(-1).to(dtype), rec)


def compress_parameter(deltas: Sequence[torch.Tensor], energy_threshold: float = 0.90,
                       max_rank: Optional[int] = None, center: bool = True, fp16: bool = True,
                       bits: int = 4, stages: int = 2, mask: Optional[torch.Tensor] = None,
                       min_mask_size: int = 10) -> Optional[Dict]:
    """
    One parameter through cli.py:317-361 (Step 4 body) and compress.py:114-170 (Step 5 body),
    masked region only.  Returns {"basis": ..., "tasks": [per-task compress_task dict],
    "recon": [per-task reconstruction of the (masked) delta]} or None when nothing to do.
    """
    if mask is not None and mask.shape == deltas[0].shape:
        if int(mask.sum()) < min_mask_size:
            return None
        vecs = [select_masked(d, mask) for d in deltas]
    else:
        vecs = [d.flatten() for d in deltas]
    if len(vecs[0]) == 0:
        return None
    basis = svd_basis(vecs, energy_threshold, max_rank, center, fp16)
    tasks, recon = [], []
    for v in vecs:
        art = compress_task(v, basis["U_high"], basis["U_low"], bits, stages, basis["mean"])
        c_low_hat = torch.from_numpy(rtvq_dequantize(art["c_low_quant"]).reshape(-1).copy())
        recon.append(reconstruct(art["c_high_fp16"].float(), c_low_hat, basis["U_high"],
                                 basis["U_low"], basis["mean"]))
        tasks.append(art)
    return {"basis": basis, "tasks": tasks, "recon": recon, "vectors": vecs}


def merge_parameter(comp: Dict, weights: Sequence[float], base: Optional[torch.Tensor] = None) -> torch.Tensor:
    """merge.py:61-141 + 144-194 (+ 429-552 with ``base``) for one parameter from compress_parameter's output:
    per task fp16 c_high -> fp32 and dequantized c_low, weights renormalised over the tasks, weighted average,
    U_high c_high + U_low c_low + mean, base + delta."""
    basis = comp["basis"]
    tot = float(sum(weights))
    w = torch.tensor([float(x) / tot for x in weights], dtype=torch.float32).view(-1, 1)
    highs = torch.stack([t["c_high_fp16"].float() for t in comp["tasks"]], dim=0)
    lows = torch.stack([torch.from_numpy(rtvq_dequantize(t["c_low_quant"]).reshape(-1).copy()) for t in comp["tasks"]],
                       dim=0)
    delta = reconstruct((highs * w).sum(dim=0), (lows * w).sum(dim=0), basis["U_high"], basis["U_low"], basis["mean"])
    return delta if base is None else base + delta


# --------------------------------------------------------------------------- synthetic inputs
def synthetic_deltas(D: int, N: int, seed: int, rank: int = 3, a: float = 0.01, eps: float = 0.002,
                     device: str = "cpu") -> List[torch.Tensor]:
    """
    SURVEY.md section 8(d) generator: delta_t = a * B (g_t * s) + eps * n_t with a decaying
    spectrum s = (1, .5, .25, ...), so that N - k >= 2 and the quantizer input is not
    degenerate.  Deterministic in (D, N, seed) on the CPU generator.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    B = torch.randn(D, rank, generator=g)
    s = torch.tensor([0.5 ** i for i in range(rank)])
    out = []
    for _ in range(N):
        gt = torch.randn(rank, generator=g)
        n = torch.randn(D, generator=g)
        out.append((a * (B @ (gt * s)) + eps * n).to(device))
    return out
