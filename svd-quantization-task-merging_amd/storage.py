"""
Artifact writer / reader with the reference's on-disk format (reference
src/svd_hybrid/storage.py:52-409, reload.py:142-238):

    <artifact_dir>/basis/<safe_name>.pt    {"masked"|"noise": {U_high, U_low, singular_values, k, mean,
                                                                energy_retained, D, N}}          CPU tensors
    <artifact_dir>/coeffs/<safe_name>.pt   {task: {"masked"|"unmasked": {c_high_fp16, c_low_quant}}}
    <artifact_dir>/diagnostics.json, config.json
    <output_dir>/merged_state_dict.pt

safe_name = name.replace("/", "_").replace("\\\\", "_") (storage.py:72).  Pure host code: tensors are moved
to the CPU and written with torch.save, exactly as the reference does; the private keys this package
attaches to basis dicts (fused-run bookkeeping) are not written.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict
from typing import Any, Dict

import torch

_BASIS_KEYS = ("U_high", "U_low", "singular_values", "k", "mean", "energy_retained", "D", "N")


def _safe(name: str) -> str:
    return name.replace("/", "_").replace("\\", "_")


def _compact(obj):
    """Deep copy in which every tensor owns exactly its own elements: the payload tensors handed out by the
    batched path are views into one packed host buffer, and torch.save writes a view's WHOLE storage."""
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _compact(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)) and not isinstance(obj, torch.Size):
        return type(obj)(_compact(v) for v in obj)
    return obj


def _basis_payload(b: Dict) -> Dict:
    return {"U_high": b["U_high"].cpu(), "U_low": b["U_low"].cpu(), "singular_values": b["singular_values"].cpu(),
            "k": b["k"], "mean": b["mean"].cpu() if b["mean"] is not None else None,
            "energy_retained": b["energy_retained"], "D": b["D"], "N": b["N"]}


def save_basis(basis: Dict, param_name: str, output_dir: str):
    """Reference storage.py:52-106."""
    d = os.path.join(output_dir, "basis")
    os.makedirs(d, exist_ok=True)
    data = {}
    if basis.get("masked") is not None:
        data["masked"] = _basis_payload(basis["masked"])
    if basis.get("noise") is not None:
        data["noise"] = _basis_payload(basis["noise"])
    torch.save(data, os.path.join(d, f"{_safe(param_name)}.pt"))


def load_basis(param_name: str, artifact_dir: str, device: str = "cpu") -> Dict:
    """Reference storage.py:109-134 (files written by save_basis hold tensors and plain scalars only)."""
    path = os.path.join(artifact_dir, "basis", f"{_safe(param_name)}.pt")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Basis file not found: {path}")
    return torch.load(path, map_location=device, weights_only=True)   # tensors, scalars, None, dict: nothing executes


def save_compressed_coefficients(compressed: Dict[str, Dict[str, Dict]], output_dir: str):
    """Reference storage.py:137-174."""
    d = os.path.join(output_dir, "coeffs")
    os.makedirs(d, exist_ok=True)
    for name, per_task in compressed.items():
        out = {}
        for task, art in per_task.items():
            a = {}
            for region in ("masked", "unmasked"):
                if art.get(region) is not None:
                    a[region] = {"c_high_fp16": _compact(art[region]["c_high_fp16"]),
                                 "c_low_quant": _compact(art[region]["c_low_quant"])}
            out[task] = a
        torch.save(out, os.path.join(d, f"{_safe(name)}.pt"))


def load_compressed_coefficients(param_name: str, artifact_dir: str, device: str = "cpu") -> Dict[str, Dict]:
    """Reference storage.py:177-202."""
    path = os.path.join(artifact_dir, "coeffs", f"{_safe(param_name)}.pt")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Coefficients file not found: {path}")
    # tensors, int / float / str / None, dict / list and torch.Size only: the weights-only unpickler accepts all of it
    return torch.load(path, map_location=device, weights_only=True)


def _serializable(obj):
    if isinstance(obj, dict):
        return {k: _serializable(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_serializable(v) for v in obj]
    if isinstance(obj, torch.Size):
        return list(obj)
    if isinstance(obj, torch.Tensor):
        return obj.cpu().tolist() if obj.numel() > 1 else obj.item()
    if hasattr(obj, "item"):
        return obj.item()
    return obj


def save_diagnostics(diagnostics: Dict[str, Any], output_dir: str):
    """Reference storage.py:205-237."""
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "diagnostics.json"), "w") as f:
        json.dump(_serializable(diagnostics), f, indent=2)


def load_diagnostics(artifact_dir: str) -> Dict[str, Any]:
    """Reference storage.py:240-258."""
    path = os.path.join(artifact_dir, "diagnostics.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Diagnostics file not found: {path}")
    with open(path) as f:
        return json.load(f)


def save_config(config, output_dir: str):
    """Reference storage.py:261-280."""
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "config.json"), "w") as f:
        d = asdict(config)
        # the key set must be exactly the reference dataclass's: its load_config does SVDHybridConfig(**json)
        # (storage.py:283-303) and raises TypeError on anything else.  The mixed-width extension is a function of
        # the parameter name and not serialisable anyway; the widths actually used are in every coeffs/*.pt
        # ("c_low_quant"["num_bits"]).
        d.pop("svd_low_bits_by_param", None)
        json.dump(d, f, indent=2)


def load_config(artifact_dir: str):
    """Reference storage.py:283-303."""
    from .config import SVDHybridConfig
    path = os.path.join(artifact_dir, "config.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Config file not found: {path}")
    with open(path) as f:
        return SVDHybridConfig(**json.load(f))


def save_all_artifacts(bases: Dict[str, Dict], compressed: Dict[str, Dict[str, Dict]], diagnostics: Dict[str, Any],
                       config, output_dir: str):
    """Reference storage.py:306-338."""
    for name, basis in bases.items():
        save_basis(basis, name, output_dir)
    save_compressed_coefficients(compressed, output_dir)
    save_diagnostics(diagnostics, output_dir)
    save_config(config, output_dir)


def load_all_artifacts(artifact_dir: str, device: str = "cpu") -> Dict[str, Any]:
    """Reference storage.py:341-389: the parameter list comes from diagnostics["per_parameter"]."""
    config = load_config(artifact_dir)
    diagnostics = load_diagnostics(artifact_dir)
    names = list(diagnostics.get("per_parameter", {}).keys())
    bases, compressed = {}, {}
    for n in names:
        try:
            bases[n] = load_basis(n, artifact_dir, device)
        except FileNotFoundError:
            pass
        try:
            compressed[n] = load_compressed_coefficients(n, artifact_dir, device)
        except FileNotFoundError:
            pass
    return {"bases": bases, "compressed": compressed, "diagnostics": diagnostics, "config": config}


def save_merged_model(merged_state_dict: Dict[str, torch.Tensor], output_dir: str,
                      filename: str = "merged_state_dict.pt"):
    """Reference storage.py:392-409."""
    os.makedirs(output_dir, exist_ok=True)
    torch.save(merged_state_dict, os.path.join(output_dir, filename))


def reconstruct_from_artifacts(artifact_dir: str, base_model_path, output_path: str = None, device: str = "cpu") -> Dict:
    """Reference reload.py:142-238: rebuild the merged model from stored artifacts.  Masks are not stored
    (reload.py:204-205), so this is exact for unmasked runs only -- as in the reference.
    ``base_model_path`` may also be an already-loaded state dict."""
    from .merge import apply_merged_deltas, merge_all_parameters
    art = load_all_artifacts(artifact_dir, device=device)
    bases, compressed, config, diagnostics = art["bases"], art["compressed"], art["config"], art["diagnostics"]
    tasks = list(diagnostics.get("task_weights", {}).keys())
    if not tasks:
        tasks = list(next(iter(compressed.values())).keys())
    weights = diagnostics.get("task_weights") or {t: 1.0 / len(tasks) for t in tasks}
    shapes = {n: torch.Size(d["original_shape"]) for n, d in diagnostics.get("per_parameter", {}).items()
              if d.get("original_shape") is not None}
    merged = merge_all_parameters(compressed, bases, {}, weights, shapes, config, device=device, verbose=False)
    if isinstance(base_model_path, dict):
        base = base_model_path
    else:
        base = torch.load(base_model_path, map_location=device, weights_only=True)
        for key in ("state_dict", "model", "model_state_dict"):   # task_vector_loader.py unwraps these
            if isinstance(base, dict) and key in base and isinstance(base[key], dict):
                base = base[key]
                break
    out = apply_merged_deltas(base, merged, device=device, verbose=False)
    if output_path:
        torch.save(out, output_path)
    return {"merged_state_dict": out, "diagnostics": diagnostics, "config": config}


def reload_merged_model_from_artifacts(artifact_dir: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Reference reload.py:60-139: a pre-saved merged_state_dict.pt in the artifact directory wins; otherwise the
    merged model is rebuilt from bases + coefficients + the base model named in the stored config."""
    pre = os.path.join(artifact_dir, "merged_state_dict.pt")
    if os.path.exists(pre):
        print(f"Loading pre-saved merged model from {pre}")
        return torch.load(pre, map_location=device, weights_only=True)
    print("No pre-saved merged model found, reconstructing from artifacts...")
    config = load_config(artifact_dir)
    base_path = getattr(config, "base_model_path", "") if not isinstance(config, dict) else config.get("base_model_path", "")
    if not base_path or not os.path.exists(base_path):
        raise FileNotFoundError(f"Base model path not found in config or doesn't exist: {base_path}. Please provide "
                                "base model path in artifacts config or use reconstruct_from_artifacts instead.")
    return reconstruct_from_artifacts(artifact_dir, base_path, None, device=device)["merged_state_dict"]
