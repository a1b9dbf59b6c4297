"""
Task clustering for cluster-weighted merging (reference
src/svd_hybrid/clustering.py:55-425).

The reference flattens every task vector into one row of a host ``[N, sum D]`` numpy matrix
(clustering.py:55-120; 24 GB at ViT-L-14 x 20) and hands it to scikit-learn.  K-means and Ward linkage
only ever look at distances between the N rows and their means, i.e. at the N x N Gram matrix of the
rows.  Here the Gram comes from one streaming pass over the task deltas where they already live
(``svdq_task_gram``: fp32 MFMA per 256-row block, fp64 across blocks, deterministic), is all-reduced
across ranks when the parameters are sharded (N*N doubles), and an exact N-dimensional embedding
``E`` with ``E E^T = Gram`` replaces the feature matrix.  scikit-learn / scipy then run on N x N
numbers with the reference's arguments (KMeans(n_clusters=k, random_state=42, n_init=10); ward).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .pipeline import CompressPlan, prepare_vector, resolve_device


def task_gram(task_vectors: Dict[str, Dict[str, torch.Tensor]], device="cuda", *, process_group=None,
              ) -> Tuple[np.ndarray, List[str]]:
    """N x N inner products (fp64, host) of the flattened task vectors, tasks in sorted order.

    Same row/column conventions as flatten_task_vectors (clustering.py:55-120): tasks sorted by name,
    parameters the union over tasks, a parameter a task lacks counts as zeros.  With ``process_group``
    (or an initialised default group and ``process_group=True``) each rank passes only the parameters it
    owns and the Gram is summed over ranks -- the one N*N all-reduce of SURVEY 8(e)."""
    dev = resolve_device(device)
    tasks = sorted(task_vectors.keys())
    names = sorted({n for tv in task_vectors.values() for n in tv.keys()})
    N = len(tasks)
    vectors, rows = [], []
    with torch.cuda.device(dev):
        for name in names:
            ref = next(tv[name] for tv in task_vectors.values() if name in tv)
            if ref.numel() == 0:
                continue
            vs = []
            for t in tasks:
                if name in task_vectors[t]:
                    vs.append(prepare_vector(task_vectors[t][name], dev))
                else:
                    vs.append(torch.zeros(ref.numel(), dtype=torch.float32, device=dev))
            vectors.append(vs)
            rows.append(ref.numel())
        if vectors:
            plan = CompressPlan(rows, N, center=False, device=dev, gram_only=True)
            G = plan.task_gram(plan.pointer_table(vectors))
        else:
            G = torch.zeros((N, N), dtype=torch.float64, device=dev)
        if process_group is not None:
            from .shard import all_reduce_gram
            all_reduce_gram(G, None if process_group is True else process_group)
        out = G.cpu().numpy()
        if vectors:
            plan.close()
    return out, tasks


def flatten_task_vectors(task_vectors: Dict[str, Dict[str, torch.Tensor]]) -> Tuple[np.ndarray, List[str]]:
    """Reference clustering.py:55-120: the [N, sum D] feature matrix (rows = tasks in sorted-name order, parameters in
    sorted-name order, zeros where a task lacks a parameter) and the task names.  Kept for API compatibility only:
    nothing in this package needs the matrix -- ``cluster_tasks`` works from the N x N Gram (``task_gram``), which is
    all the clustering ever uses of it -- and at ViT-L-14 x 20 it is 24 GB of host memory."""
    names = sorted(task_vectors.keys())
    params = sorted({p for tv in task_vectors.values() for p in tv})
    shape_of = {p: next(tv[p] for tv in task_vectors.values() if p in tv) for p in params}
    rows = []
    for t in names:
        tv = task_vectors[t]
        parts = [(tv[p] if p in tv else torch.zeros_like(shape_of[p])).flatten() for p in params]
        rows.append(torch.cat(parts, dim=0).cpu().numpy())
    return np.stack(rows, axis=0), names


def gram_embedding(G: np.ndarray) -> np.ndarray:
    """Rows e_i in R^N with e_i . e_j = G_ij (eigen-decomposition, negative round-off clipped)."""
    lam, Q = np.linalg.eigh((G + G.T) * 0.5)
    return Q * np.sqrt(np.clip(lam, 0.0, None))[None, :]


def normalized_gram(G: np.ndarray) -> np.ndarray:
    """Gram of ``features / (||features|| + 1e-8)`` (clustering.py:230-231)."""
    norm = np.sqrt(np.clip(np.diag(G), 0.0, None)) + 1e-8
    return G / norm[:, None] / norm[None, :]


def compute_kmeans_clustering(features: np.ndarray, k: int, random_state: int = 42) -> np.ndarray:
    """clustering.py:123-156."""
    from sklearn.cluster import KMeans
    if k <= 0 or k > features.shape[0]:
        raise ValueError(f"Invalid k={k} for {features.shape[0]} samples")
    return KMeans(n_clusters=k, random_state=random_state, n_init=10).fit_predict(features)


def compute_hierarchical_clustering(features: np.ndarray, k: int, method: str = "ward") -> np.ndarray:
    """clustering.py:159-195."""
    from scipy.cluster.hierarchy import fcluster, linkage
    if k <= 0 or k > features.shape[0]:
        raise ValueError(f"Invalid k={k} for {features.shape[0]} samples")
    return fcluster(linkage(features, method=method), k, criterion="maxclust") - 1


def cluster_from_gram(G: np.ndarray, task_names: List[str], k: int, method: str = "kmeans") -> Dict[str, int]:
    feats = gram_embedding(normalized_gram(G))
    if method == "kmeans":
        labels = compute_kmeans_clustering(feats, k)
    elif method == "hierarchical":
        labels = compute_hierarchical_clustering(feats, k)
    else:
        raise ValueError(f"Unknown clustering method: {method}")
    return {name: int(lab) for name, lab in zip(task_names, labels)}


def cluster_tasks(task_vectors: Dict[str, Dict[str, torch.Tensor]], k: int, method: str = "kmeans",
                  device="cuda", process_group=None) -> Dict[str, int]:
    """clustering.py:198-245: unit-normalised task vectors -> k-means / Ward labels per task name."""
    if method not in ("kmeans", "hierarchical"):
        raise ValueError(f"Unknown clustering method: {method}")
    G, tasks = task_gram(task_vectors, device, process_group=process_group)
    return cluster_from_gram(G, tasks, k, method)


def get_cluster_members(cluster_assignments: Dict[str, int]) -> Dict[int, List[str]]:
    """clustering.py:248-275."""
    clusters: Dict[int, List[str]] = {}
    for task, cid in cluster_assignments.items():
        clusters.setdefault(cid, []).append(task)
    return clusters


def cluster_statistics_from_gram(G: np.ndarray, task_names: List[str], cluster_assignments: Dict[str, int]
                                 ) -> Dict[int, Dict]:
    """||x_i - centroid||^2 = G_ii - 2 mean_j G_ij + mean_jl G_jl over the members j, l of the cluster."""
    index = {name: i for i, name in enumerate(task_names)}
    stats = {}
    for cid, members in get_cluster_members(cluster_assignments).items():
        idx = np.array([index[m] for m in members])
        sub = G[np.ix_(idx, idx)]
        d2 = np.diag(sub) - 2.0 * sub.mean(axis=1) + sub.mean()
        d = np.sqrt(np.clip(d2, 0.0, None))
        stats[cid] = {"size": len(members), "members": members,
                      "mean_distance_to_centroid": float(d.mean()),
                      "max_distance_to_centroid": float(d.max()),
                      "min_distance_to_centroid": float(d.min())}
    return stats


def compute_cluster_statistics(task_vectors: Dict[str, Dict[str, torch.Tensor]], cluster_assignments: Dict[str, int],
                               device="cuda", process_group=None) -> Dict[int, Dict]:
    """clustering.py:278-316 (distances of the UN-normalised task vectors to their cluster centroid)."""
    G, tasks = task_gram(task_vectors, device, process_group=process_group)
    return cluster_statistics_from_gram(G, tasks, cluster_assignments)


def merge_by_cluster(task_vectors: Dict[str, Dict[str, torch.Tensor]], cluster_assignments: Dict[str, int],
                     weights: Dict[str, float], device: str = "cpu") -> Dict[int, Dict[str, torch.Tensor]]:
    """clustering.py:319-371: per cluster, weighted average of the members' task vectors."""
    from .weighting import apply_weights_to_tensors
    merged = {}
    for cid, members in get_cluster_members(cluster_assignments).items():
        w = {m: weights.get(m, 1.0) for m in members}
        total = sum(w.values())
        w = {m: v / total for m, v in w.items()}
        names = {n for m in members for n in task_vectors[m].keys()}
        merged[cid] = {}
        for n in names:
            present = {m: task_vectors[m][n] for m in members if n in task_vectors[m]}
            if present:
                merged[cid][n] = apply_weights_to_tensors(present, w, device)
    return merged


def merge_cluster_results(cluster_merged: Dict[int, Dict[str, torch.Tensor]], cluster_performance: Dict[int, float],
                          device: str = "cpu") -> Dict[str, torch.Tensor]:
    """clustering.py:374-425: softmax(cluster performance) (uniform when empty) average across clusters."""
    from .weighting import apply_weights_to_tensors
    if not cluster_merged:
        return {}
    cids = list(cluster_merged.keys())
    if cluster_performance:
        share = torch.softmax(torch.tensor([cluster_performance.get(c, 1.0) for c in cids]), dim=0)
    else:
        share = torch.ones(len(cids)) / len(cids)
    w = {c: s.item() for c, s in zip(cids, share)}
    names = {n for params in cluster_merged.values() for n in params.keys()}
    out = {}
    for n in names:
        present = {c: cluster_merged[c][n] for c in cids if n in cluster_merged[c]}
        if present:
            out[n] = apply_weights_to_tensors(present, w, device)
    return out
