"""
svdq_amd -- MI355X-native SVD-Hybrid task-vector compressor (hot path of
mgradyn/SVD-Quantization-Task-Merging, src/svd_hybrid/{basis,compress,rtvq,mask_loader}.py and
quantization_utils.py), computed by hand-written gfx950 HIP kernels behind a C ABI
(include/svdq.h, libsvdq_hip.so).  Import as ``svdq_amd`` (svdq_amd.py at the repo root aliases
this directory, whose name is not a Python identifier).

The function names, argument orders, defaults and returned dict layouts are the reference's.
There is no CPU implementation in this package: a missing library raises RuntimeError.
"""
from . import _native
from .config import SVDHybridConfig
from .rtvq import (RTVQQuantizer, asymmetric_quantization, asymmetric_dequantization,
                   multistage_residual_quantization, multistage_residual_dequantization,
                   estimate_compression_ratio)
from .basis import (construct_basis, construct_masked_basis, select_rank, compute_energy_spectrum, compute_svd,
                    stack_and_center, compute_energy_statistics)
from .compress import (project_to_basis, compress_single_task, compress_masked_regions, compress_parameter,
                       compress_all_parameters)
from .mask_loader import (combine_masks, compute_union_mask, compute_intersection_mask, compute_majority_mask,
                          apply_mask_to_tensor, get_unmasked_portion, reconstruct_from_masked, load_task_masks,
                          load_single_mask, load_tall_mask_file, state_dict_to_vector, vector_to_state_dict,
                          combine_tall_masks_packed)
from .merge import (dequantize_and_average, reconstruct_from_coefficients, merge_parameter, merge_all_parameters,
                    apply_merged_deltas, merge_with_clustering)
from .weighting import (load_performance_metrics, compute_uniform_weights, compute_performance_weights,
                        compute_cluster_weights, compute_weights, apply_weights_to_tensors, get_weight_statistics)
from .clustering import (cluster_tasks, task_gram, cluster_from_gram, cluster_statistics_from_gram, flatten_task_vectors,
                         get_cluster_members, compute_cluster_statistics, merge_by_cluster, merge_cluster_results,
                         compute_kmeans_clustering, compute_hierarchical_clustering)
from .diagnostics import (compute_reconstruction_error, compute_parameter_diagnostics, compute_all_diagnostics,
                          compute_compression_statistics, print_detailed_compression_report,
                          print_diagnostics_summary, compute_coefficient_histograms)
from .storage import (save_basis, load_basis, save_compressed_coefficients, load_compressed_coefficients,
                      save_diagnostics, load_diagnostics, save_config, load_config, save_all_artifacts,
                      load_all_artifacts, save_merged_model, reconstruct_from_artifacts,
                      reload_merged_model_from_artifacts)
from .task_vector_loader import (load_checkpoint, compute_task_vector, compute_task_vectors, load_task_vectors,
                                 get_parameter_names, organize_by_parameter, flatten_task_deltas,
                                 get_task_checkpoint_paths)
from .task_vectors import TaskVector, QuantizedTaskVector, QuantizedFinetunedModel, QuantizedBaseAndTaskVector
from .ingest import ElementwiseBatch, ingest_state_dicts, quantize_state_dict, dequantize_payloads
from .driver import build_bases, run_basis_and_compress, run_basis_and_compress_from_checkpoints
from .pipeline import CompressPlan, compress_batch
from . import quantization_utils
from .quantization_utils import (absmax_quantization, dequantize_absmax, qunatization_error_check,
                                 quantization_error_check_asymmetric)
from . import cli
from . import hydra_entry
from . import torch_ops   # registers torch.ops.svdq.*

# aliases named by BASELINE.json's north_star (quantization_utils.py)
dequantize_asymmetric = asymmetric_dequantization

__all__ = [n for n in dir() if not n.startswith("_")]
