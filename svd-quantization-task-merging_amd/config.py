"""
SVDHybridConfig: the same field names, defaults and validation errors as the reference's
dataclass (reference src/svd_hybrid/config.py:157-234), so existing call sites keep working.
Only the fields the hot path reads are interpreted here; the rest are carried through.
"""
from dataclasses import dataclass, field
from typing import Callable, List, Optional


@dataclass
class SVDHybridConfig:
    svd_energy_threshold: float = 0.95
    svd_max_rank: int = 64
    svd_center: bool = True
    svd_fp16: bool = True
    svd_low_bits: int = 4
    svd_rtvq_stages: int = 2
    svd_mask_strategy: str = "union"
    svd_include_noise: bool = False
    svd_weighting: str = "uniform"
    svd_weighting_temperature: float = 5.0
    svd_cluster_k: int = 2
    svd_store_artifacts: bool = True
    svd_eval_reconstruction: bool = True
    svd_noise_shrink: float = 0.5
    svd_min_mask_size: int = 10
    svd_randomized_svd_threshold: int = 1500000
    tasks: List[str] = field(default_factory=list)
    model: str = "ViT-B-32"
    checkpoint_dir: str = ""
    base_model_path: str = ""
    mask_dir: str = ""
    performance_file: Optional[str] = None
    output_dir: str = "./svd_hybrid_output"
    artifact_dir: str = "./artifacts"
    device: str = "cuda"
    # extension (not in the reference): name -> code width, for a run that mixes widths (BASELINE config #5)
    svd_low_bits_by_param: Optional[Callable[[str], int]] = None

    def __post_init__(self):
        if self.svd_mask_strategy not in ("union", "intersection", "majority"):
            raise ValueError(f"Invalid mask strategy: {self.svd_mask_strategy}. "
                             f"Must be one of: union, intersection, majority")
        if self.svd_weighting not in ("uniform", "performance", "cluster"):
            raise ValueError(f"Invalid weighting: {self.svd_weighting}. "
                             f"Must be one of: uniform, performance, cluster")
        if self.svd_energy_threshold <= 0 or self.svd_energy_threshold > 1:
            raise ValueError(f"Energy threshold must be in (0, 1], got {self.svd_energy_threshold}")
        if self.svd_low_bits < 1 or self.svd_low_bits > 8:
            raise ValueError(f"Low bits must be in [1, 8], got {self.svd_low_bits}")
        if self.svd_rtvq_stages < 1:
            raise ValueError(f"RTVQ stages must be >= 1, got {self.svd_rtvq_stages}")
