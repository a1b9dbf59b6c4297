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
