"""
ctypes binding of libsvdq_hip.so (C ABI declared in include/svdq.h).

There is no CPU fallback: if the library is missing or fails to load, every entry point of
this package raises ``RuntimeError`` -- the HIP path is the product.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint8, c_void_p

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_PATH = os.environ.get("SVDQ_LIB_PATH") or os.path.join(_PKG_DIR, "libsvdq_hip.so")

SVDQ_OK, SVDQ_EINVAL, SVDQ_EHIP, SVDQ_EUNSUPPORTED = 0, -1, -2, -3
MASK_STRATEGIES = {"union": 0, "intersection": 1, "majority": 2}
MAX_TASKS = 32
MAX_STAGES = 8


class SvdqConfig(Structure):
    _fields_ = [("energy_threshold", c_float), ("max_rank", c_int32), ("center", c_int32),
                ("fp16", c_int32), ("low_bits", c_int32), ("rtvq_stages", c_int32),
                ("unit_rows", c_int32), ("reserved", c_int32)]


class SvdqSizes(Structure):
    _fields_ = [("workspace_bytes", c_int64), ("basis_bytes", c_int64), ("mean_floats", c_int64),
                ("small_bytes", c_int64), ("n_units", c_int32), ("n_slots", c_int32)]


class SvdqSmallLayout(Structure):
    _fields_ = [(n, c_int64) for n in ("sigma_off", "k_off", "r_off", "energy_off", "rows_off",
                                       "chigh_off", "codes_off", "scale_off", "zp_off", "rnorm_off",
                                       "coef_off", "status_off", "total_bytes")]


# name -> (restype, argtypes); mirrors include/svdq.h one to one (tests check the export list)
SIGNATURES = {
    "svdq_abi_version": (c_int32, []),
    "svdq_last_error": (c_char_p, []),
    "svdq_plan_create": (c_int32, [POINTER(c_void_p), c_int32, c_int32, POINTER(c_int64), POINTER(SvdqConfig)]),
    "svdq_plan_destroy": (None, [c_void_p]),
    "svdq_plan_set_low_bits": (c_int32, [c_void_p, POINTER(c_int32)]),
    "svdq_plan_sizes": (c_int32, [c_void_p, POINTER(SvdqSizes)]),
    "svdq_plan_small_layout": (c_int32, [c_void_p, POINTER(SvdqSmallLayout)]),
    "svdq_plan_basis_layout": (c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    "svdq_gram_center": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_eig_rank_select": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_basis_project": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_coeff_quantize": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_gram_center_range": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "svdq_eig_rank_select_range": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                             c_void_p]),
    "svdq_basis_project_range": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int32, c_int32, c_void_p]),
    "svdq_coeff_quantize_range": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "svdq_task_gram": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_ingest": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_tvq_work_bytes": (c_int64, [c_void_p]),
    "svdq_tvq_quantize": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int32, c_void_p]),
    "svdq_tvq_dequantize": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_compress": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_rtvq_work_bytes": (c_int64, [c_int64]),
    "svdq_rtvq_quantize": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "svdq_rtvq_dequantize": (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_asym16_quantize": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_asym16_dequantize": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_mask_work_bytes": (c_int64, [c_int64]),
    "svdq_mask_combine": (c_int32, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_mask_compact": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p,
                                    c_void_p]),
    "svdq_maskset_create": (c_int32, [POINTER(c_void_p), c_int32, POINTER(c_int64)]),
    "svdq_maskset_destroy": (None, [c_void_p]),
    "svdq_maskset_work_bytes": (c_int64, [c_void_p]),
    "svdq_maskset_combine": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "svdq_maskset_compact": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "svdq_maskset_indices": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_maskset_combine_indices": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_maskset_combine_packed_indices": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_maskset_count_scan": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_maskset_unit_starts": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "svdq_maskset_combine_starts": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_maskset_combine_packed_starts": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_compress_masked": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "svdq_compress_masked_from_base": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_compress_gather": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "svdq_compress_from_base": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    "svdq_compress_gather_from_base": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p]),
    "svdq_project": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    "svdq_project_work_bytes": (c_int64, [c_int64, c_int32]),
    "svdq_reconstruct": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                   c_float, c_void_p, c_void_p]),
    "svdq_recon_error_work_bytes": (c_int64, [c_int64]),
    "svdq_recon_error": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_merge_work_bytes": (c_int64, [c_void_p, c_int32]),
    "svdq_merge_coeffs": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "svdq_merge_reconstruct": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_merge": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "svdq_diagnostics_work_bytes": (c_int64, [c_void_p]),
    "svdq_diagnostics": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                   c_void_p, c_void_p]),
    "svdq_merge_masked": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                    c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "svdq_diagnostics_masked": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int32, c_void_p, c_void_p, c_void_p]),
    "svdq_mask_expand": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "svdq_hbm_probe": (c_int32, [c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into the in-tree libsvdq_hip.so, and the TORCH_LIBRARY operator shim
    into libsvdq_torch.so (csrc/Makefile)."""
    cmd = ["make", "-C", _CSRC, "-j4"]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libsvdq_hip.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"build finished but {LIB_PATH} is missing")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library. Raises loudly when it is absent: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is the only implementation of this package. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {_CSRC}`.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.svdq_abi_version() != 1:
            raise RuntimeError("libsvdq_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def last_error() -> str:
    msg = lib().svdq_last_error()
    return msg.decode() if msg else ""


def check(rc: int, what: str = "") -> None:
    """Map a return code to the exception type the reference raises for the same condition."""
    if rc == SVDQ_OK:
        return
    msg = last_error() or what
    if rc == SVDQ_EINVAL:
        raise ValueError(msg)  # the reference raises ValueError for bad arguments / empty lists
    raise RuntimeError(f"{what}: {msg} (code {rc})")
