"""
SVD basis construction with energy-based rank selection on the GPU -- same callables as the
reference's src/svd_hybrid/basis.py, computed by the Gram route in libsvdq_hip.so:

    G = Tc^T Tc (pass 1, MFMA)  ->  Jacobi eigen-solve of the N x N Gram (fp64)  ->
    sigma, k by the reference's fp32 energy rule  ->  U = Tc V Sigma^-1 (pass 2, MFMA)

Differences a caller can observe (DESIGN.md "parity statement"): singular vectors carry this
package's sign convention (largest-|v| component of each right singular vector positive), and
directions whose sigma is below 1e-6 sigma_0 (the null direction centring creates) are zero
columns where LAPACK returns an arbitrary unit vector.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _native as nat
from .pipeline import (CompressPlan, basis_dict, compress_batch, prepare_vector, resolve_device, wants_cpu)


def stack_and_center(vectors: List[torch.Tensor], center: bool = True
                     ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Reference basis.py:63-113.  Kept for API compatibility (the fused path never builds the
    [D,N] stack); plain tensor ops on the inputs' device."""
    if not vectors:
        raise ValueError("Empty vector list")
    T = torch.stack(vectors, dim=1)
    mean = None
    if center:
        mean = T.mean(dim=1, keepdim=True)
        T = T - mean
    return T, mean


def compute_energy_spectrum(singular_values: torch.Tensor) -> torch.Tensor:
    """Reference basis.py:116-156 (N scalars; host-side arithmetic in fp32 like the reference)."""
    energy = singular_values ** 2
    total = energy.sum()
    if total < 1e-10:
        return torch.ones_like(energy)
    return torch.cumsum(energy, dim=0) / total


def select_rank(singular_values: torch.Tensor, energy_threshold: float = 0.90, max_rank: Optional[int] = None,
                min_rank: int = 1) -> int:
    """Reference basis.py:159-213."""
    cum = compute_energy_spectrum(singular_values)
    k = int((cum < energy_threshold).sum().item()) + 1
    k = max(k, min_rank)
    if max_rank is not None:
        k = min(k, max_rank)
    return min(k, len(singular_values))


def _run_single(vectors: List[torch.Tensor], energy_threshold, max_rank, center, device, fp16=False,
                low_bits=4, rtvq_stages=2):
    dev = resolve_device(device)
    vs = [prepare_vector(v, dev) for v in vectors]
    D = vs[0].numel()
    for v in vs[1:]:
        if v.numel() != D:
            raise ValueError("all delta vectors must have the same length")
    return compress_batch([vs], energy_threshold=energy_threshold, max_rank=max_rank, center=center, fp16=fp16,
                          low_bits=low_bits, rtvq_stages=rtvq_stages, device=dev)


def compute_svd(matrix: torch.Tensor, full_matrices: bool = False, use_randomized: bool = False,
                random_rank: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Reference basis.py:216-249: thin SVD of a tall [D,N] matrix, results on the input's device.
    Vh comes out of the eigen-stage itself: the run's coefficient table holds u_i^T a_t = sigma_i v_i[t] in closed
    form (no centring, fp32 basis: no rounding correction on top), so Vh = diag(1/S) C^T -- N x N numbers already on
    the host, no second pass over the matrix and no library GEMM."""
    if full_matrices:
        raise ValueError("only the thin SVD (full_matrices=False) is implemented on the HIP path")
    D, N = matrix.shape
    if N > nat.MAX_TASKS or D < 1:
        raise ValueError(f"compute_svd supports 1..{nat.MAX_TASKS} columns")
    cols = [matrix[:, j].contiguous() for j in range(N)]
    plan, sm = _run_single(cols, 1.0, None, False, matrix.device if matrix.is_cuda else "cuda")
    k, r = int(sm.k[0]), int(sm.r[0])
    U_high, U_low, _ = plan.basis_tensors(0, k, r, D)
    U = torch.cat([U_high, U_low], dim=1)
    S = torch.from_numpy(sm.sigma[0, :r].copy()).to(U.device)
    C = torch.from_numpy(sm.coef[0, :N, :r].copy()).to(U.device)       # [task, direction] = sigma_i v_i[t]
    Vh = C.T / torch.where(S > 0, S, torch.ones_like(S)).unsqueeze(1)
    out_dev = matrix.device
    return U.to(out_dev), S.to(out_dev), Vh.to(out_dev)


def construct_basis(deltas: List[torch.Tensor], energy_threshold: float = 0.90, max_rank: Optional[int] = None,
                    center: bool = True, device: str = "cpu", use_randomized: bool = False,
                    verbose: bool = True) -> Dict:
    """Reference basis.py:252-409.  Returns {U_high [D,k], U_low [D,r-k], singular_values, k, mean
    [D,1]|None, energy_retained, D, N}; tensors live on ``device`` (fp32, as in the reference before the
    cli.py:354-361 cast): zero-copy views of the GPU buffers for a GPU device, copies for ``"cpu"``."""
    if not deltas:
        raise ValueError("Empty delta list")
    plan, sm = _run_single(deltas, energy_threshold, max_rank, center, device, fp16=False)
    out = basis_dict(plan, sm, 0)
    if wants_cpu(device):
        out = {key: (v.cpu() if isinstance(v, torch.Tensor) else v) for key, v in out.items()}
        plan.close()
    else:
        out["_svdq_plan"] = plan  # keeps the slab alive for the zero-copy views
    if verbose:
        print(f"   SVD basis: D={out['D']} N={out['N']} k={out['k']} energy={out['energy_retained']:.4f}")
    return out


def construct_masked_basis(masked_deltas: List[torch.Tensor], unmasked_deltas: Optional[List[torch.Tensor]],
                           energy_threshold: float = 0.90, max_rank: Optional[int] = None, center: bool = True,
                           device: str = "cpu", include_noise: bool = False, verbose: bool = False) -> Dict:
    """Reference basis.py:412-468."""
    result = {}
    if masked_deltas and len(masked_deltas[0]) > 0:
        result["masked"] = construct_basis(masked_deltas, energy_threshold=energy_threshold, max_rank=max_rank,
                                           center=center, device=device, verbose=verbose)
    else:
        result["masked"] = None
    if include_noise and unmasked_deltas and len(unmasked_deltas[0]) > 0:
        result["noise"] = construct_basis(unmasked_deltas, energy_threshold=energy_threshold, max_rank=max_rank,
                                          center=center, device=device, verbose=verbose)
    else:
        result["noise"] = None
    return result


def compute_energy_statistics(singular_values: torch.Tensor) -> Dict[str, float]:
    """Reference basis.py:471-496."""
    energy = singular_values ** 2
    total = energy.sum().item()
    n = len(singular_values)
    top = energy[0].item() if n > 0 else 0
    return {"total_energy": total, "top_singular_value": singular_values[0].item() if n > 0 else 0,
            "top_energy_ratio": (top / total if total > 0 else 0) if n > 0 else 0, "num_components": n}
