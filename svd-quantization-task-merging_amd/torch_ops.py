"""
``torch.ops.svdq.*``: the C ABI of libsvdq_hip.so exposed as PyTorch custom operators (BASELINE.json's
north_star: "exposed as PyTorch-ROCm custom ops"; SURVEY.md section 8b suggests this op set).  The operators are
registered natively -- ``TORCH_LIBRARY(svdq, ...)`` in ``csrc/svdq_torch.cpp``, built in-tree as
``libsvdq_torch.so`` and linked against ``libsvdq_hip.so`` -- so a call goes dispatcher -> C++ -> C ABI -> HIP
kernels on the current stream with no Python in between.  The schemas take and return plain tensors; there is no
ATen arithmetic and no CPU implementation behind them.

    svdq::rtvq_quantize(Tensor x, int bits, int stages) -> (Tensor codes, Tensor scale, Tensor zero_point, Tensor rnorm)
    svdq::rtvq_dequantize(Tensor codes, Tensor scale, Tensor zero_point) -> Tensor
    svdq::mask_combine(Tensor[] masks, str strategy) -> Tensor
    svdq::mask_select(Tensor x, Tensor mask, bool invert) -> Tensor
    svdq::compress(Tensor[] deltas, int n_tasks, float energy, int max_rank, bool center, bool fp16,
                   int bits, int stages) -> (Tensor small, Tensor basis, Tensor mean)
    svdq::compress_masked(Tensor[] deltas, Tensor[] masks, int n_tasks, <the six settings>) -> (small, basis, mean, rows)
    svdq::compress_gather(...same...)            the same through int32 index lists (sparse masks, N > 16)
    svdq::compress_from_base(Tensor[] finetuned, Tensor[] base, int n_tasks, <settings>) -> (small, basis, mean)
    svdq::mask_combine_indices(Tensor[] masks, int n_masks, str strategy) -> (Tensor[] combined, Tensor[] idx, Tensor counts)
    svdq::reconstruct(Tensor U_high, Tensor U_low, Tensor coef, Tensor? mean, float scale) -> Tensor
    svdq::recon_error(Tensor U_high, Tensor U_low, Tensor coef, Tensor? mean, Tensor orig) -> Tensor
    svdq::merge(Tensor small, Tensor basis, Tensor mean, int[] rows, int n_tasks, <settings>, Tensor weights,
                Tensor[] base) -> Tensor[]
    svdq::merge_masked(Tensor small, Tensor basis, Tensor mean, Tensor[] masks, int n_tasks, <settings>,
                       Tensor weights, Tensor[] base) -> Tensor[]      full-size tensors, the mask scatter fused
    svdq::diagnostics(Tensor[] deltas, Tensor[] masks, Tensor small, Tensor basis, Tensor mean, int n_tasks,
                      <settings>, bool add_mean) -> Tensor            float64 [P, n_tasks, 6]; masks [] = unmasked
    svdq::ingest(Tensor base, Tensor[] finetuned) -> Tensor[]
    svdq::task_gram(Tensor[] deltas, int n_tasks) -> Tensor
    svdq::plan_cache_size() -> int

``compress`` returns the packed buffers of a plan (layout: svdq_plan_small_layout / svdq_plan_basis_layout in
include/svdq.h); the reference-shaped dictionaries are rebuilt from them by svdq_amd.pipeline / driver.  It keeps
its plans (device tables + workspace) per (sizes, N, settings, device, stream): a repeated call with the same shapes
creates nothing and does not synchronise.  Importing this module loads the operator library once; like the HIP
library itself it is the only implementation -- a missing file is an error, not a fallback.
"""
from __future__ import annotations

import os

import torch

from . import _native as nat

OPS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsvdq_torch.so")
_loaded = False


def load() -> str:
    """Load libsvdq_torch.so (registers the ``svdq`` namespace).  The HIP library is loaded first from the path
    the ctypes layer uses, so both faces of the ABI run the same code object."""
    global _loaded
    if not _loaded:
        nat.lib()
        if not os.path.exists(OPS_LIB_PATH):
            raise RuntimeError(f"{OPS_LIB_PATH} not found: torch.ops.svdq.* is implemented there and nowhere else. "
                               f"Build it with `make -C {nat._CSRC}`.")
        torch.ops.load_library(OPS_LIB_PATH)
        _loaded = True
    return OPS_LIB_PATH


def plan_cache_size() -> int:
    """Plans currently kept by ``torch.ops.svdq.compress`` (at most 8, least recently used dropped first)."""
    return int(torch.ops.svdq.plan_cache_size())


# A tree that has not been built yet must stay importable (build() imports the package first); once the libraries
# exist, importing the package registers the operators.  Without them torch.ops.svdq has no attributes at all.
if os.path.exists(OPS_LIB_PATH) and os.path.exists(nat.LIB_PATH):
    load()
