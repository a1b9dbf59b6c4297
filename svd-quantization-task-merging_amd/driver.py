"""
Fused Step 4 + Step 5 driver (reference cli.py:311-361 and cli.py:436-442): every parameter of a
model, masked and noise regions included, goes through ONE plan -- six kernel launches and one
small D2H copy for the whole model -- and comes back as the reference's ``bases`` and
``compressed_all`` dictionaries.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from .pipeline import (BatchResult, CompressPlan, basis_dict, prepare_vector, resolve_device, task_artifact)
from . import mask_loader as ml



# the mask-walk mode (svdq_compress_masked) serves regions that hold at least this share of their tensors' elements --
# below it, skipping rows through index lists reads less than walking past them -- and the task counts whose default
# kernels walk at full speed (N <= 16; above, the walk exists but runs the one-wave pass 2, measured 20-24 % behind the
# index lists that feed the two-wave kernels)
WALK_MIN_DENSITY = 0.5
WALK_MAX_TASKS = 16


def _resident(tensors, dev) -> bool:
    """All tensors already satisfy the ABI's input contract except (possibly) for their shape: fp32, contiguous, on
    ``dev``, 16-byte aligned -- then the plan can take their addresses as they are (pointer_table re-checks)."""
    f32 = torch.float32
    for t in tensors:
        if t.dtype is not f32 or t.device != dev or not t.is_contiguous() or t.requires_grad or t.data_ptr() & 15:
            return False
    return True


def build_bases(task_vectors: Dict[str, Dict[str, torch.Tensor]], combined_masks: Optional[Dict[str, torch.Tensor]],
                config, device="cuda", base_state: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, Dict]:
    """Step 4 for all parameters.  Returns ``bases`` (reference layout: {param: {"masked": basis|None,
    "noise": basis|None}}), already cast to fp16 when ``config.svd_fp16`` (cli.py:354-361), with the
    coefficients of Step 5 attached for ``compress_all_parameters``."""
    dev = resolve_device(device)
    combined_masks = combined_masks or {}
    names = sorted({n for tv in task_vectors.values() for n in tv.keys()})
    if base_state is not None:   # compute_task_vector's eligibility (task_vector_loader.py:126-139)
        names = [n for n in names if n in base_state and base_state[n].is_floating_point()]
    tasks = list(task_vectors.keys())
    include_noise = bool(config.svd_include_noise)
    min_size = int(config.svd_min_mask_size)
    # optional partition of the parameters by code width (BASELINE config #5, "mixed 8-bit / 2-bit"): a callable
    # name -> bits on the config; the reference itself has one width per run (compress.py:180-183)
    bits_by_param = getattr(config, "svd_low_bits_by_param", None)

    # group regions by the number of tasks that have the parameter (one plan per N)
    groups: Dict[Tuple[int, str], List[dict]] = {}   # (tasks present, "plain" | "gather" | "walk") -> regions of one plan
    keep = []
    with torch.cuda.device(dev):
        masked_by_n: Dict[int, List[tuple]] = {}
        for name in names:
            present = [t for t in tasks if name in task_vectors[t]
                       and (base_state is None or task_vectors[t][name].shape == base_state[name].shape)]
            if not present:
                continue
            deltas = [task_vectors[t][name] for t in present]
            mask = combined_masks.get(name)
            if mask is not None and mask.shape == deltas[0].shape and mask.numel() > 0:
                masked_by_n.setdefault(len(present), []).append((name, present, deltas, mask))
            else:
                vs = [d if d.dim() == 1 else prepare_vector(d, dev) for d in deltas] if _resident(deltas, dev) \
                    else [prepare_vector(d, dev) for d in deltas]
                if vs[0].numel() == 0:
                    continue
                groups.setdefault((len(present), "plain"), []).append(
                    {"name": name, "region": "masked", "tasks": present, "vectors": vs, "count": None,
                     "upper": vs[0].numel(), "min": 0,
                     "base": prepare_vector(base_state[name], dev) if base_state is not None else None})
        # Masked parameters never get compacted copies of their deltas.  One count + scan per group (mask.sum() stays
        # on the device and becomes rows_dev); then, per region, one of two ways to reach the selected rows:
        #   walk   (N <= 16 and the region holds at least half of the elements): both passes walk the source rows with
        #          the mask byte beside them and compact in LDS (svdq_compress_masked) -- nothing is built per row;
        #   gather (sparse regions, N > 16): int32 index lists, rows fetched through them (svdq_compress_gather).
        for n_present, items in masked_by_n.items():
            numels = [it[3].numel() for it in items]
            ms = ml.MaskSet(numels, dev)
            mask_list = [it[3] for it in items]
            ct, cf = ms.count_scan(mask_list)
            dens = float(ct.sum().item()) / float(max(sum(numels), 1))
            can_walk = n_present <= WALK_MAX_TASKS
            walk_sig = can_walk and dens >= WALK_MIN_DENSITY
            walk_noise = can_walk and include_noise and (1.0 - dens) >= WALK_MIN_DENSITY
            it_ = if_ = None
            if not walk_sig or (include_noise and not walk_noise):
                it_, if_, _, _ = ms.indices(mask_list, want_false=include_noise and not walk_noise)
            keep.append((ms, it_, if_))
            for q, (name, present, deltas, mask_q) in enumerate(items):
                vs = [prepare_vector(d, dev) for d in deltas]
                ident = (mask_q.data_ptr(), mask_q.numel(), str(mask_q.device), mask_q.dtype)
                bvec = prepare_vector(base_state[name], dev) if base_state is not None else None
                e = {"name": name, "region": "masked", "tasks": present, "vectors": vs, "count": ct[q:q + 1],
                     "upper": vs[0].numel(), "min": min_size, "base": bvec}
                e.update(ms=ms, q=q, inv=False, mask_ident=ident)
                if not walk_sig:
                    e["index"] = it_[q]
                groups.setdefault((n_present, "walk" if walk_sig else "gather"), []).append(e)
                if include_noise:
                    e = {"name": name, "region": "noise", "tasks": present, "vectors": vs, "count": cf[q:q + 1],
                         "upper": vs[0].numel(), "min": 1, "gate": ct[q:q + 1], "base": bvec}
                    e.update(ms=ms, q=q, inv=True, mask_ident=ident)
                    if not walk_noise:
                        e["index"] = if_[q]
                    groups.setdefault((n_present, "walk" if walk_noise else "gather"), []).append(e)
        order = {n: i for i, n in enumerate(names)}
        for lst in groups.values():
            lst.sort(key=lambda e: (order[e["name"]], e["region"] != "masked"))

        bases: Dict[str, Dict] = {}
        for (n_tasks, mode), entries in groups.items():
            entries = [e for e in entries if e["upper"] > 0]
            if not entries:
                continue
            plan = CompressPlan([e["upper"] for e in entries], n_tasks,
                                energy_threshold=config.svd_energy_threshold, max_rank=config.svd_max_rank,
                                center=config.svd_center, fp16=config.svd_fp16,
                                low_bits=([int(bits_by_param(e["name"])) for e in entries] if bits_by_param
                                          else config.svd_low_bits),
                                rtvq_stages=config.svd_rtvq_stages, device=dev)
            rows_dev = None
            if any(e["count"] is not None for e in entries):
                # rows actually processed = mask.sum() (device), or 0 when below svd_min_mask_size
                parts = []
                for e in entries:
                    if e["count"] is None:
                        parts.append(torch.tensor([e["upper"]], dtype=torch.int64, device=dev))
                    else:
                        c = e["count"]
                        ok = c >= e["min"]
                        if "gate" in e:      # the noise region is only built when the signal region is (cli.py:332-338)
                            ok = ok & (e["gate"] >= min_size)
                        parts.append(torch.where(ok, c, torch.zeros_like(c)))
                rows_dev = torch.cat(parts)
            table = plan.pointer_table([e["vectors"] for e in entries])
            btab = None
            if base_state is not None:
                btab = torch.tensor([e["base"].data_ptr() for e in entries], dtype=torch.int64).to(dev)
                keep.append([e["base"] for e in entries])
            mtab = us = None
            if mode in ("walk", "gather"):
                # one source start per work unit: the walk mode's way to the rows, and what the batched consumers
                # (svdq_merge_masked / svdq_diagnostics_masked) put the merged rows back with, whichever mode compressed
                ms = entries[0]["ms"]
                mtab = torch.tensor([ms._s["mb"][e["q"]].data_ptr() for e in entries], dtype=torch.int64).to(dev)
                us = ms.unit_starts(plan, rows_dev, entry_map=[(e["q"], e["inv"]) for e in entries], mask_table=mtab)
                keep.append((mtab, us))
            if mode == "walk":
                if btab is not None:
                    plan.run_masked_from_base(table, btab, mtab, us, rows_dev)
                else:
                    plan.run_masked(table, mtab, us, rows_dev)
            elif mode == "gather":
                itab = torch.tensor([e["index"].data_ptr() for e in entries], dtype=torch.int64).to(dev)
                if btab is not None:
                    plan.run_gather_from_base(table, btab, itab, rows_dev)
                else:
                    plan.run_gather(table, itab, rows_dev)
            elif btab is not None:
                plan.run_from_base(table, btab, rows_dev)
            else:
                plan.run(table, rows_dev)
            small = plan.fetch_small()
            batch = BatchResult(plan, small, [(e["name"], e["region"]) for e in entries],
                                [e["tasks"] for e in entries])
            batch.keep = keep
            # for the batched consumers (svdq_diagnostics reads the deltas again): what the run was launched with
            batch.mode, batch.table, batch.rows_dev = mode, table, rows_dev
            batch.mask_table, batch.unit_start = mtab, us
            batch.mask_ident = {e["name"]: e["mask_ident"] for e in entries if "mask_ident" in e}
            batch.from_base = base_state is not None
            for i, e in enumerate(entries):
                slot = bases.setdefault(e["name"], {"masked": None, "noise": None})
                if int(small.rows[i]) <= 0:
                    continue
                # the views into the packed basis buffer are made when somebody looks at them
                slot[e["region"]] = LazyArtifacts((lambda pl=plan, sm=small, q=i: basis_dict(pl, sm, q)), (batch, i))
    # parameters whose masked region was skipped (mask.sum() < svd_min_mask_size) have no basis
    # entry at all in the reference (cli.py:343 guards the assignment)
    return {n: b for n, b in bases.items() if b["masked"] is not None}


class LazyArtifacts(dict):
    """A dictionary of one parameter -- {task: {"masked": art|None, "unmasked": art|None}} in ``compressed_all``, or a
    basis {U_high, U_low, singular_values, k, mean, energy_retained, D, N} in ``bases`` -- assembled from the packed
    buffers of the batched run on first access.  It IS a dict (isinstance, iteration, json, equality all behave); a caller that only
    passes the structure on (or inspects a few parameters) does not pay for assembling ~10^4 nested payload
    dictionaries per model.  storage.save_compressed_coefficients writes plain dicts (as the reference's writer
    does: it rebuilds the per-task level); pickled / torch.saved directly it comes back as a collections.OrderedDict,
    the one mapping type the weights-only unpickler admits besides dict itself."""
    __slots__ = ("_fill", "_batch", "_meta")

    def __init__(self, fill, batch=None, meta=None):
        super().__init__()
        self._fill = fill
        self._batch = batch      # (BatchResult, index): where this parameter's results live
        self._meta = meta        # compressed_all entries: which tasks / noise entry, for the batched consumers

    def _ensure(self):
        f = self._fill
        if f is not None:
            self._fill = None
            super().update(f())

    def __reduce__(self):
        from collections import OrderedDict
        self._ensure()
        return (OrderedDict, (), None, None, iter(dict.items(self)))

    def __repr__(self):
        self._ensure()
        return super().__repr__()


_MUTATORS = frozenset(("__setitem__", "__delitem__", "__ior__", "pop", "popitem", "setdefault", "update", "clear"))


def _lazy(name):
    def method(self, *a, **kw):
        self._ensure()
        for x in a:      # dict.__eq__ / __or__ / update read the OTHER operand through the C dict API, past its wrappers
            if isinstance(x, LazyArtifacts):
                x._ensure()
        if name in _MUTATORS:
            # an edited dictionary no longer describes the buffers of the fused run: the batched consumers (svdq_merge,
            # svdq_diagnostics) must not answer for it from there -- they fall back to the per-parameter route, which
            # reads the dictionary as the reference does
            self._batch = None
            self._meta = None
        return getattr(dict, name)(self, *a, **kw)
    method.__name__ = name
    return method


for _n in ("__getitem__", "__iter__", "__len__", "__contains__", "__eq__", "__ne__", "__setitem__", "__delitem__",
           "__or__", "__ror__", "__ior__", "__reversed__", "keys", "values", "items", "get", "pop", "popitem",
           "setdefault", "update", "copy", "clear", "__bool__" if hasattr(dict, "__bool__") else "__len__"):
    setattr(LazyArtifacts, _n, _lazy(_n))


def artifacts_from_batch(name: str, basis: Dict, task_vectors, config) -> Optional[Dict]:
    """{task: {"masked": art|None, "unmasked": art|None}} from the fused run, or None if this basis
    did not come from ``build_bases`` (then compress.py runs the per-task route)."""
    bm = basis.get("masked")
    if bm is None or getattr(bm, "_batch", None) is None:
        return None
    batch, i = bm._batch
    want_bits = getattr(config, "svd_low_bits_by_param", None)
    want_bits = int(want_bits(name)) if want_bits else config.svd_low_bits
    if (batch.plan.bits_of(i), batch.plan.S) != (want_bits, config.svd_rtvq_stages):
        return None
    tasks_i = batch.task_names[i]
    bn = basis.get("noise") if config.svd_include_noise else None
    have = [t for t in task_vectors.keys() if name in task_vectors[t]]

    def fill():
        out = {}
        pos = {t: j for j, t in enumerate(tasks_i)}
        npos = None
        if bn is not None and getattr(bn, "_batch", None) is not None:
            nb, j = bn._batch
            npos = {t: q for q, t in enumerate(nb.task_names[j])}
        for t in have:
            art = {"masked": None, "unmasked": None}
            if t in pos:
                art["masked"] = task_artifact(batch.plan, batch.small, i, pos[t])
            if npos is not None and t in npos:
                art["unmasked"] = task_artifact(nb.plan, nb.small, j, npos[t])
            out[t] = art
        return out

    return LazyArtifacts(fill, batch=bm._batch,
                         meta={"have": have, "tasks": tasks_i,
                               "noise": bn._batch if (bn is not None and getattr(bn, "_batch", None) is not None) else None})


def run_basis_and_compress(task_vectors, combined_masks, config, device="cuda") -> Tuple[Dict, Dict]:
    """cli.py Step 4 + Step 5 in one call: (bases, compressed_all)."""
    from .compress import compress_all_parameters
    bases = build_bases(task_vectors, combined_masks, config, device)
    return bases, compress_all_parameters(task_vectors, combined_masks or {}, bases, config, device)


def run_basis_and_compress_from_checkpoints(base_state: Dict[str, torch.Tensor],
                                            finetuned_states: Dict[str, Dict[str, torch.Tensor]], config,
                                            device="cuda", combined_masks: Optional[Dict[str, torch.Tensor]] = None
                                            ) -> Tuple[Dict, Dict]:
    """cli.py Step 1 + Step 4 + Step 5 without materialising the task vectors: ``finetuned - base`` is formed
    inside the two streaming passes (svdq_compress_from_base; with ``combined_masks`` the masked parameters go
    through svdq_compress_gather_from_base).  Same (bases, compressed_all) as load_task_vectors +
    run_basis_and_compress, bit for bit."""
    from .compress import compress_all_parameters
    bases = build_bases(finetuned_states, combined_masks, config, device, base_state=base_state)
    return bases, compress_all_parameters(finetuned_states, combined_masks or {}, bases, config, device)
