"""
Hydra-style entry (reference src/svd_hybrid/hydra_entry.py:66-118): build an ``SVDHybridConfig`` from a nested
configuration and run the pipeline.  Hydra / OmegaConf are not needed for that: anything with ``.get`` -- a
``DictConfig`` or a plain ``dict`` of the same shape (top level: tasks, paths, device; ``method``: the ``svd_*``
settings) -- is accepted.  The defaults applied for missing keys are the entry point's own
(hydra_entry.py:79-101: energy 0.90, max rank 128, temperature 1.0), which differ from the dataclass's.
"""
from __future__ import annotations

from typing import Any, Dict

from .config import SVDHybridConfig

# (field, section, default when the key is absent): "top" = the configuration itself, "method" = its method section
_FIELDS = (
    ("tasks", "top", []), ("checkpoint_dir", "top", ""), ("base_model_path", "top", ""), ("mask_dir", "top", ""),
    ("output_dir", "top", "./svd_hybrid_output"), ("artifact_dir", "top", "./artifacts"), ("device", "top", "cuda"),
    ("svd_energy_threshold", "method", 0.90), ("svd_max_rank", "method", 128), ("svd_center", "method", True),
    ("svd_fp16", "method", True), ("svd_low_bits", "method", 4), ("svd_rtvq_stages", "method", 2),
    ("svd_mask_strategy", "method", "union"), ("svd_include_noise", "method", False),
    ("svd_noise_shrink", "method", 0.5), ("svd_weighting", "method", "uniform"), ("performance_file", "method", None),
    ("svd_weighting_temperature", "method", 1.0), ("svd_cluster_k", "method", 2),
    ("svd_store_artifacts", "method", True), ("svd_eval_reconstruction", "method", True),
)


def config_from_hydra(cfg) -> SVDHybridConfig:
    method = cfg.get("method", {})
    if not hasattr(method, "get"):          # `method: svd_hybrid` given as a bare name: no overrides
        method = {}
    values: Dict[str, Any] = {}
    for name, section, default in _FIELDS:
        src = cfg if section == "top" else method
        v = src.get(name, default)
        values[name] = list(v) if name == "tasks" else v
    return SVDHybridConfig(**values)


def run_from_hydra(cfg):
    from .cli import run_svd_hybrid_pipeline
    return run_svd_hybrid_pipeline(config_from_hydra(cfg))


def main(cfg=None):
    """With Hydra installed the reference decorates this with ``@hydra.main``; here it takes the configuration object
    (or nothing: then it says how to call it, as the reference's fallback does)."""
    if cfg is None:
        print("Error: pass a configuration (dict or DictConfig) to run_from_hydra / main; "
              "the command-line entry is scripts/run_svd_hybrid.py")
        return None
    return run_from_hydra(cfg)
