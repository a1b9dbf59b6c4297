"""
Task-vector loading with the reference's callables (reference
src/svd_hybrid/task_vector_loader.py:56-291).  ``finetuned - base`` runs on the GPU: per task through
``compute_task_vector`` (one launch per call) or for all tasks at once through ``load_task_vectors`` /
``compute_task_vectors`` (one launch per model: the base model is read once for the N tasks and the
deltas land in the buffers the compressor's pointer table then names).

Checkpoints are read with ``torch.load(..., weights_only=True)``: the reference passes
``weights_only=False`` (task_vector_loader.py:82), i.e. it unpickles arbitrary objects; a checkpoint that
is a pickled ``nn.Module`` is therefore refused here -- save its ``state_dict()`` instead.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

from .ingest import ingest_state_dicts


def _extract_state_dict(checkpoint) -> Dict[str, torch.Tensor]:
    if isinstance(checkpoint, torch.nn.Module):
        return checkpoint.state_dict()
    if isinstance(checkpoint, dict):
        for key in ("state_dict", "model", "model_state_dict"):
            if key in checkpoint:
                return checkpoint[key]
    return checkpoint


def load_checkpoint(checkpoint_path: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """task_vector_loader.py:56-98 (safe loader, see module docstring)."""
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"Checkpoint not found: {checkpoint_path}")
    return _extract_state_dict(torch.load(checkpoint_path, map_location=device, weights_only=True))


def compute_task_vector(base_state: Dict[str, torch.Tensor], finetuned_state: Dict[str, torch.Tensor],
                        device: str = "cpu") -> Dict[str, torch.Tensor]:
    """task_vector_loader.py:103-141: {param: finetuned - base} for keys in both with equal shapes."""
    return ingest_state_dicts(base_state, {"_": finetuned_state}, device)["_"]


def compute_task_vectors(base_state: Dict[str, torch.Tensor], finetuned_states: Dict[str, Dict[str, torch.Tensor]],
                         device: str = "cuda") -> Dict[str, Dict[str, torch.Tensor]]:
    """All tasks in one pass over the base model (the batched form of compute_task_vector)."""
    return ingest_state_dicts(base_state, finetuned_states, device)


def load_task_vectors(base_model_path: str, task_checkpoint_paths: Dict[str, str], device: str = "cpu",
                      filter_keys: Optional[List[str]] = None) -> Dict[str, Dict[str, torch.Tensor]]:
    """task_vector_loader.py:144-189: checkpoints are staged on the host, the subtraction is one GPU pass."""
    def keep(state):
        if filter_keys is None:
            return state
        return {k: v for k, v in state.items() if any(pat in k for pat in filter_keys)}

    print(f"Loading base model from {base_model_path}")
    base_state = keep(load_checkpoint(base_model_path, "cpu"))
    finetuned = {}
    for task, path in task_checkpoint_paths.items():
        print(f"Loading task vector for {task} from {path}")
        finetuned[task] = keep(load_checkpoint(path, "cpu"))
    return ingest_state_dicts(base_state, finetuned, device)


def get_parameter_names(task_vectors: Dict[str, Dict[str, torch.Tensor]]) -> List[str]:
    """task_vector_loader.py:192-206."""
    return sorted({name for tv in task_vectors.values() for name in tv.keys()})


def organize_by_parameter(task_vectors: Dict[str, Dict[str, torch.Tensor]]) -> Dict[str, Dict[str, torch.Tensor]]:
    """task_vector_loader.py:209-229."""
    by_param: Dict[str, Dict[str, torch.Tensor]] = {}
    for task, tv in task_vectors.items():
        for name, delta in tv.items():
            by_param.setdefault(name, {})[task] = delta
    return by_param


def flatten_task_deltas(task_vectors: Dict[str, Dict[str, torch.Tensor]], param_name: str
                        ) -> Tuple[List[torch.Tensor], List[str]]:
    """task_vector_loader.py:232-255."""
    deltas, names = [], []
    for task, tv in task_vectors.items():
        if param_name in tv:
            deltas.append(tv[param_name].flatten())
            names.append(task)
    return deltas, names


def get_task_checkpoint_paths(checkpoint_dir: str, task_names: List[str]) -> Dict[str, str]:
    """task_vector_loader.py:258-291: first existing of the reference's five naming patterns."""
    paths = {}
    for task in task_names:
        for cand in (f"{task}.pt", f"{task}.pth", os.path.join(task, "checkpoint.pt"), os.path.join(task, "model.pt"),
                     os.path.join(task, "finetuned.pt")):
            full = os.path.join(checkpoint_dir, cand)
            if os.path.exists(full):
                paths[task] = full
                break
        else:
            raise FileNotFoundError(f"No checkpoint found for task {task} in {checkpoint_dir}")
    return paths
