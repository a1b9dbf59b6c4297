"""
Batched elementwise front end of the hot path: task-vector ingest
(``finetuned - base`` for N tasks in one pass) and whole-tensor quantization ("TVQ",
quantization_utils.py:60-172) of ragged tensor lists, on the same unit tables the compressor uses.
Everything here is a thin host layer over ``svdq_ingest`` / ``svdq_tvq_quantize`` /
``svdq_tvq_dequantize`` (csrc/svdq_ingest.hip): a handful of launches for a whole state dict x N tasks
instead of a Python loop of tiny torch ops per parameter.
"""
from __future__ import annotations

from ctypes import c_void_p
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _native as nat
from .pipeline import CompressPlan, _ptr, _stream_ptr, prepare_vector, resolve_device

METHODS = {"asymmetric": 0, "absmax": 1}


def _table(tensors: Sequence[torch.Tensor], dev: torch.device) -> torch.Tensor:
    return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64).to(dev)


def _method_id(method: str) -> int:
    # the reference treats every method other than "asymmetric" as absmax (task_vectors.py:700-760)
    return 0 if method == "asymmetric" else 1


class ElementwiseBatch:
    """P ragged tensors x N "tasks" sharing one plan (unit table) for the elementwise kernels."""

    def __init__(self, rows: Sequence[int], n_tasks: int, device):
        self.dev = resolve_device(device)
        self.rows = [int(r) for r in rows]
        self.P, self.N = len(self.rows), int(n_tasks)
        self.plan = CompressPlan(self.rows, self.N, center=False, device=self.dev, gram_only=True)
        self.lib = nat.lib()
        with torch.cuda.device(self.dev):
            self.work = torch.empty(int(self.lib.svdq_tvq_work_bytes(self.plan._h)), dtype=torch.uint8, device=self.dev)

    def close(self):
        self.plan.close()

    # flat lists are parameter-major: index p * N + t
    def ingest(self, base: Sequence[torch.Tensor], finetuned: Sequence[torch.Tensor], with_stats: bool = False
               ) -> List[torch.Tensor]:
        dev = self.dev
        with torch.cuda.device(dev):
            deltas = [torch.empty(self.rows[i // self.N], dtype=torch.float32, device=dev) for i in range(self.P * self.N)]
            tb, tf, td = _table(base, dev), _table(finetuned, dev), _table(deltas, dev)
            nat.check(self.lib.svdq_ingest(self.plan._h, _ptr(tb), _ptr(tf), _ptr(td),
                                           _ptr(self.work) if with_stats else c_void_p(0), _stream_ptr()), "svdq_ingest")
        self._keep = (base, finetuned, tb, tf, td)
        return deltas

    def quantize(self, xs: Sequence[torch.Tensor], bits: int, method: str, stats_ready: bool = False
                 ) -> Tuple[List[torch.Tensor], torch.Tensor, Optional[torch.Tensor]]:
        dev, mode = self.dev, _method_id(method)
        with torch.cuda.device(dev):
            cdt = torch.uint8 if mode == 0 else (torch.int16 if int(bits) == 16 else torch.int8)
            codes = [torch.empty(self.rows[i // self.N], dtype=cdt, device=dev) for i in range(self.P * self.N)]
            scale = torch.empty(self.P * self.N, dtype=torch.float32, device=dev)
            zp = torch.empty(self.P * self.N, dtype=torch.float32, device=dev) if mode == 0 else None
            tx, tc = _table(xs, dev), _table(codes, dev)
            nat.check(self.lib.svdq_tvq_quantize(self.plan._h, _ptr(tx), mode, int(bits), _ptr(tc), _ptr(scale), _ptr(zp),
                                                 _ptr(self.work), int(stats_ready), _stream_ptr()), "svdq_tvq_quantize")
        self._keep = (xs, tx, tc)
        return codes, scale, zp

    def dequantize(self, codes: Sequence[torch.Tensor], scale: torch.Tensor, zp: Optional[torch.Tensor], method: str,
                   add: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
        dev, mode = self.dev, _method_id(method)
        if mode == 1 and codes and codes[0].dtype == torch.int16:
            mode = 2          # absmax with int16 codes (qbit = 16)
        with torch.cuda.device(dev):
            outs = [torch.empty(self.rows[i // self.N], dtype=torch.float32, device=dev) for i in range(self.P * self.N)]
            tc, to = _table(codes, dev), _table(outs, dev)
            ta = _table(add, dev) if add is not None else None
            nat.check(self.lib.svdq_tvq_dequantize(self.plan._h, _ptr(tc), mode, _ptr(scale), _ptr(zp), _ptr(ta), _ptr(to),
                                                   _stream_ptr()), "svdq_tvq_dequantize")
        self._keep = (codes, add, tc, to, ta)
        return outs


def _codes_flat(q: torch.Tensor, dev: torch.device, mode: int) -> torch.Tensor:
    want = torch.uint8 if mode == 0 else (torch.int16 if q.dtype == torch.int16 else torch.int8)
    c = q.detach().to(device=dev)
    if c.dtype != want:
        c = c.to(want)
    return c.contiguous().view(-1)


def ingest_state_dicts(base_state: Dict[str, torch.Tensor], finetuned_states: Dict[str, Dict[str, torch.Tensor]],
                       device="cuda", skip_int64: bool = False, skip_uint8: bool = False
                       ) -> Dict[str, Dict[str, torch.Tensor]]:
    """{task: {param: finetuned - base}} for every task at once.  Parameter eligibility per task follows
    compute_task_vector (task_vector_loader.py:126-139): key present in both, shapes equal; iteration in
    the base state dict's order.  Parameters are grouped by the set of tasks that have them (normally one
    group) and each group is one kernel launch."""
    dev = resolve_device(device)
    tasks = list(finetuned_states.keys())
    out: Dict[str, Dict[str, torch.Tensor]] = {t: {} for t in tasks}
    groups: Dict[Tuple[str, ...], List[str]] = {}
    for key, b in base_state.items():
        if not isinstance(b, torch.Tensor):
            continue
        if (skip_int64 and b.dtype == torch.int64) or (skip_uint8 and b.dtype == torch.uint8):
            continue
        present = []
        for t in tasks:
            f = finetuned_states[t].get(key)
            if f is None:
                continue
            if f.shape != b.shape:
                print(f"Warning: Shape mismatch for {key}, skipping")
                continue
            present.append(t)
        if present:
            groups.setdefault(tuple(present), []).append(key)
    for present, keys in groups.items():
        for c0 in range(0, len(present), nat.MAX_TASKS):
            chunk = present[c0:c0 + nat.MAX_TASKS]
            live = [k for k in keys if base_state[k].numel() > 0]
            for k in keys:
                if base_state[k].numel() == 0:
                    for t in chunk:
                        out[t][k] = torch.zeros(base_state[k].shape, dtype=torch.float32, device=dev)
            if not live:
                continue
            with torch.cuda.device(dev):
                base = [prepare_vector(base_state[k], dev) for k in live]
                ft = [prepare_vector(finetuned_states[t][k], dev) for k in live for t in chunk]
                batch = ElementwiseBatch([b.numel() for b in base], len(chunk), dev)
                deltas = batch.ingest(base, ft)
                torch.cuda.current_stream().synchronize()   # inputs may be temporaries of prepare_vector
                batch.close()
            for i, k in enumerate(live):
                for j, t in enumerate(chunk):
                    out[t][k] = deltas[i * len(chunk) + j].view(base_state[k].shape)
    # keep the base state dict's key order inside every task vector
    order = {k: i for i, k in enumerate(base_state.keys())}
    return {t: dict(sorted(tv.items(), key=lambda kv: order[kv[0]])) for t, tv in out.items()}


def quantize_state_dict(state: Dict[str, torch.Tensor], qbit: int = 8, method: str = "asymmetric", device="cuda",
                        skip_int64: bool = True, skip_uint8: bool = True) -> Dict[str, Dict]:
    """{param: {"quantized", "scale", ["zero_point"], "shape"}} for a whole state dict in three launches
    (the per-parameter loop of QuantizedFinetunedModel.__init__, task_vectors.py:764-845)."""
    dev = resolve_device(device)
    mode = _method_id(method)
    if not ((1 <= int(qbit) <= 8 and not (mode == 1 and int(qbit) < 2)) or (mode == 1 and int(qbit) == 16)):
        raise NotImplementedError(f"qbit={qbit}: codes of 8 bits or fewer (and int16 for absmax) are implemented")
    keys = []
    for k, v in state.items():
        if not isinstance(v, torch.Tensor):
            continue
        if (skip_int64 and v.dtype == torch.int64) or (skip_uint8 and v.dtype == torch.uint8):
            continue
        keys.append(k)
    live = [k for k in keys if state[k].numel() > 0]
    out: Dict[str, Dict] = {}
    if live:
        with torch.cuda.device(dev):
            xs = [prepare_vector(state[k], dev) for k in live]
            batch = ElementwiseBatch([x.numel() for x in xs], 1, dev)
            codes, scale, zp = batch.quantize(xs, qbit, method)
            torch.cuda.current_stream().synchronize()
            batch.close()
        res = {}
        for i, k in enumerate(live):
            payload = {"quantized": codes[i].view(state[k].shape), "scale": scale[i]}
            if mode == 0:
                payload["zero_point"] = zp[i]
            payload["shape"] = state[k].shape
            res[k] = payload
    else:
        res = {}
    for k in keys:   # preserve the input order
        if k in res:
            out[k] = res[k]
        else:
            raise RuntimeError(f"cannot quantize empty tensor {k!r} (min() of an empty tensor)")
    return out


def dequantize_payloads(payloads: Dict[str, Dict], method: str = "asymmetric", device="cuda",
                        add: Optional[Dict[str, torch.Tensor]] = None, reshape: bool = True) -> Dict[str, torch.Tensor]:
    """Inverse of quantize_state_dict for every parameter in one launch; ``add[param]`` (when present for a
    parameter) is added in the same pass."""
    dev = resolve_device(device)
    mode = _method_id(method)
    keys = list(payloads.keys())
    if not keys:
        return {}
    out: Dict[str, torch.Tensor] = {}
    for with_add, wide in ((False, False), (True, False), (False, True), (True, True)):
        sel = [k for k in keys if ((add is not None and k in add) == with_add) and payloads[k]["quantized"].numel() > 0
               and (payloads[k]["quantized"].dtype == torch.int16) == wide]
        if not sel:
            continue
        with torch.cuda.device(dev):
            codes = [_codes_flat(payloads[k]["quantized"], dev, mode) for k in sel]
            scale = torch.stack([torch.as_tensor(payloads[k]["scale"]).to(dev).float().reshape(()) for k in sel])
            zp = None
            if mode == 0:
                zp = torch.stack([torch.as_tensor(payloads[k].get("zero_point", 0.0)).to(dev).float().reshape(())
                                  for k in sel])
            addl = [prepare_vector(add[k], dev) for k in sel] if with_add else None
            batch = ElementwiseBatch([c.numel() for c in codes], 1, dev)
            vals = batch.dequantize(codes, scale, zp, method, addl)
            torch.cuda.current_stream().synchronize()
            batch.close()
        for k, v in zip(sel, vals):
            shape = payloads[k].get("shape", payloads[k]["quantized"].shape) if reshape else payloads[k]["quantized"].shape
            out[k] = v.view(shape)
    for k in keys:
        if k not in out:
            out[k] = torch.zeros(payloads[k]["quantized"].shape, dtype=torch.float32, device=dev)
    return {k: out[k] for k in keys}
