"""Shared helpers for the test-suite (fixture loading, sign alignment, comparisons)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def bits_equal(a, b):
    """Bit-for-bit equality of two float arrays (NaN == NaN, -0.0 != +0.0)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    w = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    av, bv = a.view(w), b.view(w)
    if np.array_equal(av, bv):
        return True
    nan_both = np.isnan(a) & np.isnan(b)
    return bool(np.all((av == bv) | nan_both))


def align_signs(U, U_ref):
    """Per-column sign s_j = sign(<u_j, uref_j>); returns (U * s, s). Singular vectors are
    defined up to sign (and up to rotation inside equal-sigma clusters, which callers avoid)."""
    U = np.asarray(U, dtype=np.float64)
    U_ref = np.asarray(U_ref, dtype=np.float64)
    s = np.sign((U * U_ref).sum(axis=0))
    s[s == 0] = 1.0
    return U * s, s


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def as_tensors(arr2d):
    return [torch.from_numpy(np.ascontiguousarray(r)) for r in arr2d]


def _tree(o):
    """Structure of a saved object: key trees, tensor dtypes / shapes, python types (no values)."""
    if isinstance(o, torch.Tensor):
        return {"tensor": str(o.dtype).replace("torch.", ""), "shape": list(o.shape), "device": o.device.type}
    if isinstance(o, torch.Size):
        return {"torch.Size": list(o)}
    if isinstance(o, dict):
        return {"dict": {str(k): _tree(v) for k, v in o.items()}}
    if isinstance(o, (list, tuple)):
        return {type(o).__name__: [_tree(v) for v in o]}
    return type(o).__name__


def _json_tree(o):
    if isinstance(o, dict):
        return {k: _json_tree(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_json_tree(v) for v in o]
    return type(o).__name__


def artifact_manifest(root):
    """Same definition as tests/golden/make_golden.py::artifact_manifest (which applied it to the files the
    reference's writer produced): relative file names, and per file the key tree with dtypes and shapes (.pt, loaded
    weights-only) or the key tree with value types (.json)."""
    import json
    man = {}
    for dirpath, _, files in os.walk(root):
        for f in sorted(files):
            full = os.path.join(dirpath, f)
            rel = os.path.relpath(full, root).replace(os.sep, "/")
            if f.endswith(".pt"):
                man[rel] = _tree(torch.load(full, map_location="cpu", weights_only=True))
            elif f.endswith(".json"):
                man[rel] = _json_tree(json.load(open(full)))
    return man


def diag_fp32_bound(U_high, U_low, c_high, c_low, x, mean=None):
    """Forward-error bound of the fp32 arithmetic in ``U_high.float() @ c_high + U_low.float() @ c_low`` (reference
    diagnostics.py:210-212) for each of the six error numbers: every element of the reconstruction carries at most
    ``(r + 2) * 2^-24 * (|U| |c|)_i`` of rounding whatever order an fp32 implementation sums the r products in (the
    reference's BLAS, a chain of fmas, an MFMA), and the error vector inherits it.  Measured on tests/golden/diag.npz:
    the reference's OWN fp32 numbers sit up to 7e-5 (max_absolute_error, uncentred runs whose error is ~1e-3 of the
    tensor) from the fp64 evaluation of the same formula on the same stored numbers -- inside this bound, far outside
    a flat 1e-5.  Returns {key: absolute tolerance against the fp64 evaluation}."""
    import torch
    A = torch.cat([U_high.double().abs(), U_low.double().abs()], dim=1) @ torch.cat(
        [c_high.double().abs().reshape(-1), c_low.double().abs().reshape(-1)])
    r = U_high.shape[1] + U_low.shape[1]
    if mean is not None:      # the add_mean extension rounds once more, at the size of |rec| + |mean|
        A = A + mean.double().abs().reshape(-1)
        r += 1
    g = (r + 2) * 2.0 ** -24
    xn = float(x.double().norm())
    l2, linf, l1 = float(A.norm()) * g, float(A.max()) * g if A.numel() else 0.0, float(A.mean()) * g if A.numel() else 0.0
    return {"absolute_error": l2, "relative_error": l2 / xn if xn > 1e-10 else 0.0, "max_absolute_error": linf,
            "mean_absolute_error": l1, "original_norm": 0.0, "reconstructed_norm": l2}


def diag_check(got, x, U_high, U_low, c_high, c_low, what="", mean=None, slack=1.0):
    """All six numbers of one (parameter, task) against diagnostics.py:186-215 evaluated in fp64 on the SAME stored
    artifacts (no basis freedom): relative 2e-6 (the fp32 conversion of the norms the kernels report) plus the fp32
    bound above."""
    from oracle import svd_hybrid_oracle as orc
    import torch
    want = orc.parameter_task_diagnostics(x, U_high, U_low, c_high, c_low, dtype=torch.float64, mean=mean)
    tol = diag_fp32_bound(U_high, U_low, c_high, c_low, x, mean=mean)
    for key in orc.DIAG_KEYS:
        g, w = float(got[key]), float(want[key])
        assert abs(g - w) <= 2e-6 * abs(w) + slack * tol[key] + 1e-30, (what, key, g, w, tol[key])


def diag_check_chunked(got, x, U_high, U_low, c_high, c_low, what="", chunk=1 << 24):
    """diag_check for tensors too long for one library call (2^30 rows: a single fp64 matrix-vector product of that
    length is itself suspect): the same formula and the same bound, accumulated over row chunks in fp64."""
    import math
    import torch
    c = torch.cat([c_high.double().reshape(-1), c_low.double().reshape(-1)])
    ca = c.abs()
    D = x.numel()
    se = sx = sr = sa = mx = a2 = a1 = amax = 0.0
    for lo in range(0, D, chunk):
        sl = slice(lo, min(lo + chunk, D))
        U = torch.cat([U_high[sl].double(), U_low[sl].double()], dim=1)
        rec = (U * c).sum(dim=1)      # elementwise: a library matrix-vector product refuses / mishandles this many rows
        xs = x[sl].double().reshape(-1)
        e = xs - rec
        A = (U.abs() * ca).sum(dim=1)
        se += float((e * e).sum()); sx += float((xs * xs).sum()); sr += float((rec * rec).sum())
        sa += float(e.abs().sum()); mx = max(mx, float(e.abs().max()))
        a2 += float((A * A).sum()); a1 += float(A.sum()); amax = max(amax, float(A.max()))
    g = (U_high.shape[1] + U_low.shape[1] + 2) * 2.0 ** -24
    xn = math.sqrt(sx)
    want = {"absolute_error": math.sqrt(se), "relative_error": math.sqrt(se) / xn if xn > 1e-10 else 0.0,
            "max_absolute_error": mx, "mean_absolute_error": sa / D, "original_norm": xn, "reconstructed_norm": math.sqrt(sr)}
    tol = {"absolute_error": math.sqrt(a2) * g, "relative_error": math.sqrt(a2) * g / xn if xn > 1e-10 else 0.0,
           "max_absolute_error": amax * g, "mean_absolute_error": a1 / D * g, "original_norm": 0.0,
           "reconstructed_norm": math.sqrt(a2) * g}
    for key, w in want.items():
        assert abs(float(got[key]) - w) <= 2e-6 * abs(w) + tol[key] + 1e-30, (what, key, float(got[key]), w, tol[key])
