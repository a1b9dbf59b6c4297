"""
Task weights for the coefficient-space merge (reference
src/svd_hybrid/weighting.py:58-396).  Host-sized work: N scalars.  Same callables, argument order,
defaults and fallbacks ("no file -> uniform with a warning", unknown task -> 1.0) as the reference.
"""
from __future__ import annotations

import json
import math
from typing import Dict, List, Optional

import torch


def _norm_name(name: str) -> str:
    return name.lower().replace("_", "").replace("-", "")


def load_performance_metrics(performance_file: str, task_names: List[str]) -> Dict[str, float]:
    """weighting.py:61-117: exact key first, then case/underscore/dash-insensitive match, else 1.0."""
    with open(performance_file, "r") as f:
        table = json.load(f)
    loose = {}
    for key, value in table.items():
        loose.setdefault(_norm_name(key), value)  # first matching key wins, as the reference's scan does
    metrics = {}
    for task in task_names:
        if task in table:
            metrics[task] = float(table[task])
        elif _norm_name(task) in loose:
            metrics[task] = float(loose[_norm_name(task)])
        else:
            print(f"Warning: No performance metric found for task {task}, using default 1.0")
            metrics[task] = 1.0
    return metrics


def compute_uniform_weights(task_names: List[str]) -> Dict[str, float]:
    """weighting.py:120-139."""
    w = 1.0 / len(task_names)
    return {name: w for name in task_names}


def compute_performance_weights(performance_metrics: Dict[str, float], temperature: float = 1.0) -> Dict[str, float]:
    """weighting.py:142-191: fp32 softmax(acc / T), as torch computes it."""
    if not performance_metrics:
        return {}
    names = list(performance_metrics.keys())
    acc = torch.tensor([performance_metrics[n] for n in names])
    w = torch.softmax(acc / temperature, dim=0)
    return {n: x.item() for n, x in zip(names, w)}


def compute_cluster_weights(task_names: List[str], cluster_assignments: Dict[str, int],
                            cluster_performance: Optional[Dict[int, float]] = None) -> Dict[str, float]:
    """weighting.py:194-263: cluster share (uniform or softmax of performance) split evenly over members."""
    members: Dict[int, int] = {}
    for task in task_names:
        cid = cluster_assignments.get(task, 0)
        members[cid] = members.get(cid, 0) + 1
    if cluster_performance is not None:
        cids = list(members.keys())
        share_t = torch.softmax(torch.tensor([cluster_performance.get(c, 1.0) for c in cids]), dim=0)
        share = {c: s.item() for c, s in zip(cids, share_t)}
    else:
        share = {c: 1.0 / len(members) for c in members}
    raw = {}
    for task in task_names:
        cid = cluster_assignments.get(task, 0)
        raw[task] = share[cid] / members[cid]
    total = sum(raw.values())
    return {k: v / total for k, v in raw.items()} if total > 0 else raw


def compute_weights(task_names: List[str], weighting_strategy: str = "uniform",
                    performance_file: Optional[str] = None, temperature: float = 1.0,
                    cluster_assignments: Optional[Dict[str, int]] = None,
                    cluster_performance: Optional[Dict[int, float]] = None) -> Dict[str, float]:
    """weighting.py:266-329."""
    if weighting_strategy == "uniform":
        return compute_uniform_weights(task_names)
    if weighting_strategy == "performance":
        if performance_file is None:
            print("Warning: No performance file provided, using uniform weights")
            return compute_uniform_weights(task_names)
        return compute_performance_weights(load_performance_metrics(performance_file, task_names), temperature)
    if weighting_strategy == "cluster":
        if cluster_assignments is None:
            print("Warning: No cluster assignments provided, using uniform weights")
            return compute_uniform_weights(task_names)
        return compute_cluster_weights(task_names, cluster_assignments, cluster_performance)
    raise ValueError(f"Unknown weighting strategy: {weighting_strategy}")


def apply_weights_to_tensors(tensors: Dict, weights: Dict, device: str = "cpu") -> torch.Tensor:
    """weighting.py:332-372: weighted average over sorted keys, weights renormalised to sum 1.

    Computed on the GPU like everything else in this package (``device="cpu"`` is the reference's default
    argument; there is no CPU path here, and without a GPU this raises)."""
    from .pipeline import resolve_device
    if not tensors:
        raise ValueError("Empty tensor dictionary")
    names = sorted(tensors.keys())
    first = tensors[names[0]]
    dev = first.device if first.is_cuda else resolve_device(device)
    stack = torch.stack([tensors[n].to(dev).float() for n in names], dim=0)
    w = torch.tensor([weights.get(n, 1.0 / len(names)) for n in names], device=dev, dtype=torch.float32)
    w = (w / w.sum()).view([len(names)] + [1] * (stack.dim() - 1))
    return (stack * w).sum(dim=0)


def get_weight_statistics(weights: Dict[str, float]) -> Dict[str, float]:
    """weighting.py:375-396 (population std, entropy with the reference's +1e-10 guard)."""
    if not weights:
        return {}
    v = list(weights.values())
    mean = sum(v) / len(v)
    return {"min": min(v), "max": max(v), "mean": mean,
            "std": math.sqrt(sum((x - mean) ** 2 for x in v) / len(v)),
            "entropy": -sum(x * math.log(x + 1e-10) for x in v)}
