"""
``torch.ops.svdq.*``: the C ABI of libsvdq_hip.so exposed as PyTorch custom operators (BASELINE.json's
north_star: "exposed as PyTorch-ROCm custom ops"; SURVEY.md section 8b suggests this op set).  The schemas
take and return plain tensors; every op enqueues hand-written HIP kernels on the current stream through
``include/svdq.h`` -- there is no ATen arithmetic and no CPU implementation behind them.

    svdq::rtvq_quantize(Tensor x, int bits, int stages) -> (Tensor codes, Tensor scale, Tensor zero_point, Tensor rnorm)
    svdq::rtvq_dequantize(Tensor codes, Tensor scale, Tensor zero_point) -> Tensor
    svdq::mask_combine(Tensor[] masks, str strategy) -> Tensor
    svdq::mask_select(Tensor x, Tensor mask, bool invert) -> Tensor
    svdq::compress(Tensor[] deltas, int n_tasks, float energy, int max_rank, bool center, bool fp16,
                   int bits, int stages) -> (Tensor small, Tensor basis, Tensor mean)
    svdq::ingest(Tensor base, Tensor[] finetuned) -> Tensor[]
    svdq::task_gram(Tensor[] deltas, int n_tasks) -> Tensor

``compress`` returns the packed buffers of a plan (layout: svdq_plan_small_layout / svdq_plan_basis_layout in
include/svdq.h); the reference-shaped dictionaries are rebuilt from them by svdq_amd.pipeline / driver.
Importing this module registers the operators once.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import _native as nat
from .pipeline import CompressPlan, prepare_vector, resolve_device

_LIB = torch.library.Library("svdq", "DEF")
_LIB.define("rtvq_quantize(Tensor x, int bits, int stages) -> (Tensor, Tensor, Tensor, Tensor)")
_LIB.define("rtvq_dequantize(Tensor codes, Tensor scale, Tensor zero_point) -> Tensor")
_LIB.define("mask_combine(Tensor[] masks, str strategy) -> Tensor")
_LIB.define("mask_select(Tensor x, Tensor mask, bool invert) -> Tensor")
_LIB.define("compress(Tensor[] deltas, int n_tasks, float energy, int max_rank, bool center, bool fp16, int bits, "
            "int stages) -> (Tensor, Tensor, Tensor)")
_LIB.define("ingest(Tensor base, Tensor[] finetuned) -> Tensor[]")
_LIB.define("task_gram(Tensor[] deltas, int n_tasks) -> Tensor")


def _rtvq_quantize(x: torch.Tensor, bits: int, stages: int):
    from .rtvq import _quantize_device
    codes, scale, zp, rnorm = _quantize_device(prepare_vector(x, resolve_device(x.device)), bits, stages)
    return codes.contiguous(), scale, zp, rnorm


def _rtvq_dequantize(codes: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor) -> torch.Tensor:
    from .rtvq import _dequantize
    if codes.dim() == 1:
        codes, scale, zero_point = codes[None], scale.reshape(1), zero_point.reshape(1)
    return _dequantize([codes[s] for s in range(codes.shape[0])], list(scale), list(zero_point), codes.device)


def _mask_combine(masks: List[torch.Tensor], strategy: str) -> torch.Tensor:
    from .mask_loader import _combine
    return _combine(list(masks), strategy)


def _mask_select(x: torch.Tensor, mask: torch.Tensor, invert: bool) -> torch.Tensor:
    from .mask_loader import _select
    return _select(x, mask, invert)


# Plans (device tables + workspace) are kept per (sizes, N, settings, device, stream): a repeated call with the same
# shapes creates nothing and does not synchronise -- its kernels are ordered behind the previous call's on the same
# stream, which is also what makes sharing the workspace safe.  Output buffers are fresh per call (the caller owns them).
_PLAN_CACHE: "dict[tuple, CompressPlan]" = {}
_PLAN_CACHE_MAX = 8


def _cached_plan(rows, n_tasks, energy, max_rank, center, fp16, bits, stages, dev) -> CompressPlan:
    key = (tuple(rows), n_tasks, float(energy), int(max_rank), bool(center), bool(fp16), int(bits), int(stages),
           dev.index, torch.cuda.current_stream(dev).cuda_stream)
    plan = _PLAN_CACHE.pop(key, None)
    if plan is None:
        plan = CompressPlan(rows, n_tasks, energy_threshold=energy, max_rank=max_rank if max_rank > 0 else None,
                            center=center, fp16=fp16, low_bits=bits, rtvq_stages=stages, device=dev)
        while len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
            old = _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
            torch.cuda.synchronize(dev)                 # its workspace may still be in use
            old.close()
    else:
        plan.fresh_outputs()
    _PLAN_CACHE[key] = plan                              # most recently used last
    return plan


def _compress(deltas: List[torch.Tensor], n_tasks: int, energy: float, max_rank: int, center: bool, fp16: bool,
              bits: int, stages: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if n_tasks < 1 or len(deltas) % n_tasks != 0 or not deltas:
        raise ValueError("deltas must hold n_tasks tensors per parameter (parameter-major)")
    dev = resolve_device(deltas[0].device)
    P = len(deltas) // n_tasks
    vecs = [[prepare_vector(deltas[p * n_tasks + t], dev) for t in range(n_tasks)] for p in range(P)]
    plan = _cached_plan([v[0].numel() for v in vecs], n_tasks, energy, max_rank, center, fp16, bits, stages, dev)
    plan.run(plan.pointer_table(vecs))
    plan._keep = None      # inputs are only read by the kernels just enqueued; temporaries made by prepare_vector are
    #                        released stream-ordered by the caching allocator (same stream), so nothing has to be held
    mean = plan.mean if plan.mean is not None else torch.empty(0, dtype=torch.float32, device=dev)
    return plan.small, plan.basis, mean


def _ingest(base: torch.Tensor, finetuned: List[torch.Tensor]) -> List[torch.Tensor]:
    from .ingest import ElementwiseBatch
    dev = resolve_device(base.device)
    b = prepare_vector(base, dev)
    batch = ElementwiseBatch([b.numel()], len(finetuned), dev)
    out = batch.ingest([b], [prepare_vector(f, dev) for f in finetuned])
    torch.cuda.current_stream(dev).synchronize()
    batch.close()
    return [o.view(base.shape) for o in out]


def _task_gram(deltas: List[torch.Tensor], n_tasks: int) -> torch.Tensor:
    dev = resolve_device(deltas[0].device)
    P = len(deltas) // n_tasks
    vecs = [[prepare_vector(deltas[p * n_tasks + t], dev) for t in range(n_tasks)] for p in range(P)]
    plan = CompressPlan([v[0].numel() for v in vecs], n_tasks, center=False, device=dev, gram_only=True)
    G = plan.task_gram(plan.pointer_table(vecs))
    torch.cuda.current_stream(dev).synchronize()
    plan.close()
    return G


for _name, _fn in (("rtvq_quantize", _rtvq_quantize), ("rtvq_dequantize", _rtvq_dequantize),
                   ("mask_combine", _mask_combine), ("mask_select", _mask_select), ("compress", _compress),
                   ("ingest", _ingest), ("task_gram", _task_gram)):
    _LIB.impl(_name, _fn, "CUDA")          # "CUDA" is the HIP device key on ROCm builds of PyTorch
