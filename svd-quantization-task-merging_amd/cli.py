"""
End-to-end driver with the reference's command line (reference src/svd_hybrid/cli.py:73-961,
scripts/run_svd_hybrid.py): load checkpoints -> task vectors -> masks -> bases + compression -> weights ->
merge -> diagnostics -> artifacts.  Same flags, defaults, output files (merged_state_dict.pt, weights.json,
clusters.json, basis/ coeffs/ diagnostics.json config.json) and returned dict; every stage runs through the
HIP entry points of this package:

  Step 1  load_task_vectors           one subtraction pass for all tasks (svdq_ingest)
  Step 2  load_task_masks + combine_masks   batched mask combine (svdq_maskset_combine)
  Step 4+5 run_basis_and_compress     the hot path: six kernel launches for the whole model (svdq_compress)
  Step 6  compute_weights / cluster_tasks   N x N task Gram on the GPU (svdq_task_gram)
  Step 7  merge_all_parameters / merge_with_clustering, apply_merged_deltas (svdq_reconstruct, svdq_mask_expand)
  Step 8  compute_all_diagnostics     fused reconstruction error (svdq_recon_error)

There is no CPU path: without a GPU the driver raises (the reference silently falls back to "cpu", cli.py:135).
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict

import torch

from .clustering import cluster_tasks, get_cluster_members
from .config import SVDHybridConfig
from .diagnostics import (compute_all_diagnostics, compute_compression_statistics, print_detailed_compression_report,
                          print_diagnostics_summary)
from .driver import run_basis_and_compress, run_basis_and_compress_from_checkpoints
from .mask_loader import combine_masks, combine_tall_masks_packed, load_task_masks
from .merge import apply_merged_deltas, merge_all_parameters, merge_with_clustering
from .pipeline import resolve_device
from .storage import save_all_artifacts, save_merged_model
from .task_vector_loader import (get_parameter_names, get_task_checkpoint_paths, load_checkpoint, load_task_vectors)
from .weighting import compute_weights


def _find_tall_mask_file(mask_dir: str, n_tasks: int):
    """The reference's TALL mask file names (mask_loader.py:303-320), plus the .npz / .pt containers this package reads."""
    for stem in (f"TALL_mask_{n_tasks}task", f"TALL_mask_{n_tasks}tasks", f"tall_mask_{n_tasks}task",
                 f"tall_mask_{n_tasks}tasks"):
        for ext in (".npy", ".npz", ".pt"):
            path = os.path.join(mask_dir, stem + ext)
            if os.path.exists(path):
                return path
    return None


def run_svd_hybrid_pipeline(config: SVDHybridConfig) -> Dict:
    """cli.py:73-778.  Returns {"merged_state_dict", "diagnostics", "bases", "compressed"}."""
    device = str(resolve_device(config.device))
    print(f"[1/8] loading base model and {len(config.tasks)} task checkpoints")
    base_state_dict = load_checkpoint(config.base_model_path, device="cpu")
    paths = get_task_checkpoint_paths(config.checkpoint_dir, config.tasks)
    has_masks = bool(config.mask_dir) and os.path.exists(config.mask_dir)
    # Nothing downstream needs the task vectors themselves when there is no clustering and no reconstruction
    # diagnostics: then finetuned - base is formed inside the two streaming passes (svdq_compress_from_base; masked
    # parameters through svdq_compress_gather_from_base) and the deltas are never materialised.
    from_checkpoints = config.svd_weighting != "cluster" and not config.svd_eval_reconstruction
    if from_checkpoints:
        float_base = {k: v for k, v in base_state_dict.items() if isinstance(v, torch.Tensor) and v.is_floating_point()}
        task_vectors = {}
        for t, pth in paths.items():
            print(f"Loading fine-tuned weights for {t} from {pth}")
            sd = load_checkpoint(pth, device="cpu")
            task_vectors[t] = {k: v for k, v in sd.items() if k in float_base and v.shape == float_base[k].shape}
    else:
        task_vectors = load_task_vectors(config.base_model_path, paths, device=device)
        # the merge is defined on floating-point parameters; integer buffers (step counters ...) are carried over
        task_vectors = {t: {k: v for k, v in tv.items() if base_state_dict[k].is_floating_point()}
                        for t, tv in task_vectors.items()}

    combined_masks: Dict[str, torch.Tensor] = {}
    if config.mask_dir and os.path.exists(config.mask_dir):
        print(f"[2/8] loading masks from {config.mask_dir} (strategy: {config.svd_mask_strategy})")
        tall = _find_tall_mask_file(config.mask_dir, len(config.tasks))
        if tall is not None:     # bit-packed tall masks: combined on the GPU straight from the packed streams
            float_ref = {k: v for k, v in base_state_dict.items() if isinstance(v, torch.Tensor)}
            combined_masks = combine_tall_masks_packed(tall, config.tasks, float_ref, config.svd_mask_strategy, device)
        else:
            task_masks = load_task_masks(config.mask_dir, config.tasks, device=device,
                                         reference_state_dict=base_state_dict)
            combined_masks = combine_masks(task_masks, strategy=config.svd_mask_strategy, device=device, verbose=False)
    else:
        print("[2/8] no masks")

    param_names = get_parameter_names(task_vectors)
    original_shapes = {}
    for tv in task_vectors.values():
        for name, delta in tv.items():
            original_shapes.setdefault(name, delta.shape)
    print(f"[3/8] {len(param_names)} parameters, {sum(s.numel() for s in original_shapes.values()):,} elements")

    print("[4-5/8] bases + compression (svdq_compress)")
    if from_checkpoints:   # task_vectors holds the fine-tuned weights here (same names and shapes as the deltas)
        bases, compressed_all = run_basis_and_compress_from_checkpoints(float_base, task_vectors, config, device,
                                                                        combined_masks=combined_masks)
    else:
        bases, compressed_all = run_basis_and_compress(task_vectors, combined_masks, config, device)
    stats = compute_compression_statistics(task_vectors, compressed_all, bases, config)
    print_detailed_compression_report(stats, config)

    print(f"[6/8] weights ({config.svd_weighting})")
    cluster_assignments = None
    if config.svd_weighting == "cluster":
        cluster_assignments = cluster_tasks(task_vectors, config.svd_cluster_k, method="kmeans", device=device)
        for cid, members in get_cluster_members(cluster_assignments).items():
            print(f"      cluster {cid}: {members}")
    weights = compute_weights(config.tasks, weighting_strategy=config.svd_weighting,
                              performance_file=config.performance_file, temperature=config.svd_weighting_temperature,
                              cluster_assignments=cluster_assignments)

    print("[7/8] merge")
    if config.svd_weighting == "cluster" and cluster_assignments is not None:
        merged_deltas = merge_with_clustering(compressed_all, bases, combined_masks, weights, cluster_assignments,
                                              original_shapes, config, device=device)
    else:
        merged_deltas = merge_all_parameters(compressed_all, bases, combined_masks, weights, original_shapes, config,
                                             device=device, verbose=False)
    merged_state_dict = apply_merged_deltas({k: v.clone() for k, v in base_state_dict.items()}, merged_deltas,
                                            device=device, verbose=False)

    if config.svd_eval_reconstruction:
        print("[8/8] diagnostics")
        diagnostics = compute_all_diagnostics(task_vectors, compressed_all, bases, combined_masks, config, device=device)
        diagnostics["task_weights"] = weights
        if cluster_assignments:
            diagnostics["cluster_assignments"] = cluster_assignments
        print_diagnostics_summary(diagnostics)
    else:
        diagnostics = {"task_weights": weights}

    if config.svd_store_artifacts:
        save_all_artifacts(bases, compressed_all, diagnostics, config, config.artifact_dir)
    save_merged_model(merged_state_dict, config.output_dir)
    os.makedirs(config.output_dir, exist_ok=True)
    with open(os.path.join(config.output_dir, "weights.json"), "w") as f:
        json.dump(weights, f, indent=2)
    if cluster_assignments:
        with open(os.path.join(config.output_dir, "clusters.json"), "w") as f:
            json.dump(cluster_assignments, f, indent=2)
    return {"merged_state_dict": merged_state_dict, "diagnostics": diagnostics, "bases": bases,
            "compressed": compressed_all, "compression_statistics": stats}


def run_svd_hybrid(config: SVDHybridConfig) -> Dict:
    """Reference run.py:40-65."""
    return run_svd_hybrid_pipeline(config)


def parse_args(argv=None):
    """cli.py:781-875: the reference's flags and defaults."""
    p = argparse.ArgumentParser(description="SVD-Hybrid merging method combining Tall Masks and TVQ (MI355X HIP path)")
    p.add_argument("--config", type=str, default=None, help="Path to JSON config file (overrides command-line args)")
    p.add_argument("--quantize-config", type=str, default=None, help="Path to quantization config JSON")
    p.add_argument("--load-config", type=str, default=None, help="Path to loading config JSON")
    p.add_argument("--tasks", nargs="+", help="List of task identifiers")
    p.add_argument("--model", type=str, default="ViT-B-32", help="Model identifier (e.g., ViT-B-32)")
    p.add_argument("--checkpoint-dir", type=str, help="Directory containing task checkpoints")
    p.add_argument("--base-model-path", type=str, help="Path to base model checkpoint")
    p.add_argument("--mask-dir", type=str, default="", help="Directory containing tall masks")
    p.add_argument("--load-tv-type", type=str, default=None,
                   choices=["standard", "quantized", "quantized_finetuned", "quantized_base_and_tv"],
                   help="Type of task vector to load")
    p.add_argument("--load-task-bits", type=int, default=8, help="Bits for task vector quantization when loading")
    p.add_argument("--load-base-bits", type=int, default=8, help="Bits for base model quantization when loading")
    p.add_argument("--energy-threshold", type=float, default=0.95, help="Energy retention threshold for rank selection")
    p.add_argument("--max-rank", type=int, default=64, help="Maximum rank cap")
    p.add_argument("--center", action="store_true", default=True, help="Center task matrix before SVD")
    p.add_argument("--no-center", action="store_false", dest="center", help="Don't center task matrix")
    p.add_argument("--fp16", action="store_true", default=True, help="Use FP16 for bases")
    p.add_argument("--no-fp16", action="store_false", dest="fp16", help="Use FP32 for bases")
    p.add_argument("--low-bits", type=int, default=4, help="Bits for low-energy coefficient quantization")
    p.add_argument("--rtvq-stages", type=int, default=2, help="Number of RTVQ refinement stages")
    p.add_argument("--mask-strategy", type=str, default="union", choices=["union", "intersection", "majority"],
                   help="Mask combination strategy")
    p.add_argument("--include-noise", action="store_true", help="Process unmasked (noise) region")
    p.add_argument("--noise-shrink", type=float, default=0.5, help="Shrinkage factor for noise region")
    p.add_argument("--weighting", type=str, default="uniform", choices=["uniform", "performance", "cluster"],
                   help="Task weighting strategy")
    p.add_argument("--performance-file", type=str, default=None, help="Path to performance metrics JSON file")
    p.add_argument("--temperature", type=float, default=5.0, help="Temperature for performance-based weighting")
    p.add_argument("--cluster-k", type=int, default=2, help="Number of clusters for cluster-based weighting")
    p.add_argument("--store-artifacts", action="store_true", help="Store compression artifacts")
    p.add_argument("--eval-reconstruction", action="store_true", default=True, help="Evaluate reconstruction error")
    p.add_argument("--no-eval-reconstruction", action="store_false", dest="eval_reconstruction",
                   help="Skip reconstruction evaluation")
    p.add_argument("--output-dir", type=str, default="./svd_hybrid_output", help="Output directory for merged model")
    p.add_argument("--artifact-dir", type=str, default="./artifacts", help="Directory for artifact storage")
    p.add_argument("--device", type=str, default="cuda", help="Device to use")
    return p.parse_args(argv)


def main(argv=None):
    """cli.py:878-961: JSON config files override the flags for tasks / checkpoint_dir / base_model_path."""
    args = parse_args(argv)
    over = {}
    if args.config:
        with open(args.config) as f:
            over = json.load(f)
    for path, section in ((args.quantize_config, "quantization"), (args.load_config, "loading")):
        if path:
            with open(path) as f:
                cfg = json.load(f)
            over.update(cfg.get(section, {}))
            if "tasks" in cfg and not args.tasks:
                over["tasks"] = cfg["tasks"]
            over.update(cfg.get("checkpoints", {}))
    tasks = over.get("tasks", args.tasks)
    checkpoint_dir = over.get("checkpoint_dir", args.checkpoint_dir)
    base_model_path = over.get("base_model_path", args.base_model_path)
    if not tasks:
        raise ValueError("--tasks must be specified either via command-line or config file")
    if not checkpoint_dir:
        raise ValueError("--checkpoint-dir must be specified either via command-line or config file")
    if not base_model_path:
        raise ValueError("--base-model-path must be specified either via command-line or config file")
    config = SVDHybridConfig(
        tasks=tasks, model=args.model, checkpoint_dir=checkpoint_dir, base_model_path=base_model_path,
        mask_dir=args.mask_dir, svd_energy_threshold=args.energy_threshold, svd_max_rank=args.max_rank,
        svd_center=args.center, svd_fp16=args.fp16, svd_low_bits=args.low_bits, svd_rtvq_stages=args.rtvq_stages,
        svd_mask_strategy=args.mask_strategy, svd_include_noise=args.include_noise, svd_noise_shrink=args.noise_shrink,
        svd_weighting=args.weighting, performance_file=args.performance_file, svd_weighting_temperature=args.temperature,
        svd_cluster_k=args.cluster_k, svd_store_artifacts=args.store_artifacts,
        svd_eval_reconstruction=args.eval_reconstruction, output_dir=args.output_dir, artifact_dir=args.artifact_dir,
        device=args.device)
    return run_svd_hybrid_pipeline(config)


if __name__ == "__main__":
    main()
