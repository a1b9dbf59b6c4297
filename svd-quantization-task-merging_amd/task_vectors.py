"""
TaskVector and the quantized task-vector family on the GPU (reference
task_vectors.py:61-1010 with quantization_utils.py:60-172).  Same classes, constructor arguments and
payload dictionaries ({"quantized", "scale", "zero_point", "shape"}); the arithmetic runs in the batched
kernels of csrc/svdq_ingest.hip (a few launches per state dict) and the tensors live on the GPU.

Differences a user can observe: checkpoints given as paths are read with ``weights_only=True`` (the
reference unpickles with ``weights_only=False``); deltas and dequantized tensors are fp32; ``qbit = 16``
is not implemented; ``verbose`` prints a two-line summary instead of the reference's tutorial text.
``dequantize_absmax`` multiplies by the scale exactly as the reference does (quantization_utils.py:102-134).
"""
from __future__ import annotations

from typing import Dict, Optional, Union

import torch

from .ingest import dequantize_payloads, ingest_state_dicts, quantize_state_dict
from .task_vector_loader import _extract_state_dict

Checkpoint = Union[str, Dict[str, torch.Tensor]]


def _state(checkpoint: Checkpoint) -> Dict[str, torch.Tensor]:
    if isinstance(checkpoint, str):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
    return _extract_state_dict(checkpoint)


class TaskVector:
    """task_vectors.py:98-635: ``vector[param] = finetuned[param] - pretrained[param]``."""

    def __init__(self, pretrained_checkpoint: Checkpoint, finetuned_checkpoint: Checkpoint,
                 task_name: Optional[str] = None, skip_int64: bool = True, skip_uint8: bool = True,
                 verbose: bool = True, device="cuda"):
        self.task_name = task_name
        self.verbose = verbose
        pre, fin = _state(pretrained_checkpoint), _state(finetuned_checkpoint)
        self.vector = ingest_state_dicts(pre, {"_": fin}, device, skip_int64=skip_int64, skip_uint8=skip_uint8)["_"]
        if verbose:
            print(f"TaskVector{' ' + task_name if task_name else ''}: {len(self.vector)} parameters, "
                  f"{len(pre) - len(self.vector)} skipped")

    @classmethod
    def from_many(cls, pretrained_checkpoint: Checkpoint, finetuned_checkpoints: Dict[str, Checkpoint],
                  skip_int64: bool = True, skip_uint8: bool = True, device="cuda") -> Dict[str, "TaskVector"]:
        """All tasks in ONE pass over the pretrained model (base read once for the N tasks)."""
        pre = _state(pretrained_checkpoint)
        vecs = ingest_state_dicts(pre, {t: _state(c) for t, c in finetuned_checkpoints.items()}, device,
                                  skip_int64=skip_int64, skip_uint8=skip_uint8)
        out = {}
        for t, v in vecs.items():
            tv = cls.__new__(cls)
            tv.task_name, tv.verbose, tv.vector = t, False, v
            out[t] = tv
        return out

    def _derived(self, name, vector) -> "TaskVector":
        res = TaskVector.__new__(TaskVector)
        res.task_name, res.verbose, res.vector = name, False, vector
        return res

    def __add__(self, other: "TaskVector") -> "TaskVector":
        name = f"{self.task_name}+{other.task_name}" if self.task_name and other.task_name else None
        return self._derived(name, {k: v + other.vector[k] for k, v in self.vector.items() if k in other.vector})

    def __sub__(self, other: "TaskVector") -> "TaskVector":
        name = f"{self.task_name}-{other.task_name}" if self.task_name and other.task_name else None
        return self._derived(name, {k: v - other.vector[k] for k, v in self.vector.items() if k in other.vector})

    def __mul__(self, scalar: float) -> "TaskVector":
        return self._derived(self.task_name, {k: v * scalar for k, v in self.vector.items()})

    def __rmul__(self, scalar: float) -> "TaskVector":
        return self.__mul__(scalar)

    def apply_to(self, pretrained_checkpoint: Checkpoint, verbose: bool = None) -> Dict[str, torch.Tensor]:
        """pretrained + vector for the keys the vector has; other entries are returned unchanged."""
        pre = _state(pretrained_checkpoint)
        out = {}
        for k, v in pre.items():
            if k in self.vector:
                d = self.vector[k]
                out[k] = v.to(d.device) + d
            else:
                out[k] = v.clone() if isinstance(v, torch.Tensor) else v
        return out


class QuantizedTaskVector:
    """task_vectors.py:638-761."""

    def __init__(self, quantized_deltas: Dict[str, Dict], method: str = "asymmetric", device="cuda"):
        self.quantized_deltas = quantized_deltas
        self.method = method
        self.device = device

    @classmethod
    def from_task_vector(cls, task_vector: Union[TaskVector, Dict[str, torch.Tensor]], qbit: int = 8,
                         method: str = "asymmetric", device="cuda") -> "QuantizedTaskVector":
        vec = task_vector.vector if isinstance(task_vector, TaskVector) else task_vector
        return cls(quantize_state_dict(vec, qbit, method, device, skip_int64=False, skip_uint8=False), method, device)

    def dequantize(self) -> Dict[str, torch.Tensor]:
        # the reference returns the flat-or-shaped tensor the codes have; no reshape to payload["shape"] here
        return dequantize_payloads(self.quantized_deltas, self.method, self.device, reshape=False)

    def apply_to(self, pretrained_checkpoint: Checkpoint) -> Dict[str, torch.Tensor]:
        """pretrained + dequantized delta, fused in the dequantization pass."""
        pre = _state(pretrained_checkpoint)
        both = {k: pre[k] for k in self.quantized_deltas if k in pre}
        merged = dequantize_payloads({k: self.quantized_deltas[k] for k in both}, self.method, self.device, add=both,
                                     reshape=False)
        return {k: (merged[k].view(v.shape) if k in merged else v) for k, v in pre.items()}


class QuantizedFinetunedModel:
    """task_vectors.py:764-874: quantize the fine-tuned weights themselves."""

    def __init__(self, finetuned_checkpoint: Checkpoint, qbit: int = 8, method: str = "asymmetric",
                 skip_int64: bool = True, skip_uint8: bool = True, device="cuda"):
        self.qbit, self.method, self.device = qbit, method, device
        self.quantized_weights = quantize_state_dict(_state(finetuned_checkpoint), qbit, method, device,
                                                     skip_int64=skip_int64, skip_uint8=skip_uint8)

    def dequantize(self) -> Dict[str, torch.Tensor]:
        return dequantize_payloads(self.quantized_weights, self.method, self.device)

    def get_task_vector(self, pretrained_checkpoint: Checkpoint) -> Dict[str, torch.Tensor]:
        pre = _state(pretrained_checkpoint)
        fin = self.dequantize()
        keys = [k for k in fin if k in pre]
        # fin - pre on the GPU; iterate in the dequantized model's order like the reference
        vec = ingest_state_dicts({k: pre[k] for k in keys}, {"_": {k: fin[k] for k in keys}}, self.device)["_"]
        return {k: vec[k] for k in keys if k in vec}


class QuantizedBaseAndTaskVector:
    """task_vectors.py:877-1010: base model and task vector quantized separately (different bit widths)."""

    def __init__(self, pretrained_checkpoint: Checkpoint, task_vector: Union[TaskVector, Dict[str, torch.Tensor]],
                 base_qbit: int = 8, task_qbit: int = 8, method: str = "asymmetric", skip_int64: bool = True,
                 skip_uint8: bool = True, device="cuda"):
        vec = task_vector.vector if isinstance(task_vector, TaskVector) else task_vector
        self.method, self.base_qbit, self.task_qbit, self.device = method, base_qbit, task_qbit, device
        self.quantized_base = quantize_state_dict(_state(pretrained_checkpoint), base_qbit, method, device,
                                                  skip_int64=skip_int64, skip_uint8=skip_uint8)
        self.quantized_task = quantize_state_dict(vec, task_qbit, method, device, skip_int64=False, skip_uint8=False)

    def dequantize(self) -> Dict[str, torch.Tensor]:
        out = dequantize_payloads(self.quantized_base, self.method, self.device)
        task = dequantize_payloads(self.quantized_task, self.method, self.device,
                                   add={k: out[k] for k in self.quantized_task if k in out})
        for k, v in task.items():
            out[k] = v
        return out
