"""
Diagnostics with the reference's dict layout and numbers (reference
src/svd_hybrid/diagnostics.py:72-321).  The per-task reconstruction error is one fused streaming kernel
(`svdq_recon_error`: reconstruct and compare without materialising the reconstruction).

Quirk kept on purpose: like the reference, ``compute_parameter_diagnostics`` reconstructs
WITHOUT adding the mean back (diagnostics.py:210-212) and compares against the uncentred original, so
``mean_relative_error`` is large whenever ``svd_center`` is on.  A "fixed" number would be a behaviour change.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import _native as nat
from .pipeline import prepare_vector, resolve_device, _ptr, _stream_ptr
from .rtvq import RTVQQuantizer, estimate_compression_ratio

_KEYS = ("absolute_error", "relative_error", "max_absolute_error", "mean_absolute_error", "original_norm",
         "reconstructed_norm")


def _metrics(out6: torch.Tensor) -> Dict[str, float]:
    vals = out6.cpu().tolist()
    return {k: float(v) for k, v in zip(_KEYS, vals)}


def compute_reconstruction_error(original_delta: torch.Tensor, reconstructed_delta: torch.Tensor) -> Dict[str, float]:
    """Reference diagnostics.py:72-117."""
    lib = nat.lib()
    dev = resolve_device(original_delta.device if original_delta.is_cuda else
                         (reconstructed_delta.device if reconstructed_delta.is_cuda else "cuda"))
    x = prepare_vector(original_delta, dev)
    r = prepare_vector(reconstructed_delta, dev)
    if x.numel() != r.numel():
        raise ValueError(f"Shape mismatch: original {tuple(original_delta.shape)} vs reconstructed "
                         f"{tuple(reconstructed_delta.shape)}")
    n = x.numel()
    if n == 0:
        return {k: 0.0 for k in _KEYS}
    out = torch.empty(6, dtype=torch.float64, device=dev)
    work = torch.empty(int(lib.svdq_recon_error_work_bytes(n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_recon_error(None, None, 0, n, 0, 0, None, None, _ptr(r), _ptr(x), _ptr(out), _ptr(work),
                                       _stream_ptr()), "svdq_recon_error")
    return _metrics(out)


def _fused_error(x: torch.Tensor, U_high, U_low, c_high, c_low, dev) -> Dict[str, float]:
    lib = nat.lib()
    D = x.numel()
    k = U_high.shape[1] if U_high.dim() == 2 else 0
    nl = U_low.shape[1] if U_low.dim() == 2 else 0
    fp16 = (U_high.dtype == torch.float16) if k else (U_low.dtype == torch.float16)
    dt = torch.float16 if fp16 else torch.float32
    uh = U_high.to(device=dev, dtype=dt).contiguous() if k else None
    ul = U_low.to(device=dev, dtype=dt).contiguous() if nl else None
    coef = torch.cat([c_high.to(dev).float().reshape(-1), c_low.to(dev).float().reshape(-1)]).contiguous()
    out = torch.empty(6, dtype=torch.float64, device=dev)
    work = torch.empty(int(lib.svdq_recon_error_work_bytes(D)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(lib.svdq_recon_error(_ptr(uh), _ptr(ul), int(fp16), D, k, nl, _ptr(coef), None, None, _ptr(x),
                                       _ptr(out), _ptr(work), _stream_ptr()), "svdq_recon_error")
    return _metrics(out)


def compute_parameter_diagnostics(param_name: str, task_vectors: Dict[str, Dict[str, torch.Tensor]],
                                  compressed_params: Dict[str, Dict], basis: Dict, mask: Optional[torch.Tensor],
                                  quantizer: RTVQQuantizer, device: str = "cpu") -> Dict:
    """Reference diagnostics.py:120-231 (same keys, same order of operations, Q1 included)."""
    from .mask_loader import apply_mask_to_tensor
    dev = resolve_device(device)
    diag = {"param_name": param_name, "original_shape": None, "masked_size": 0, "unmasked_size": 0,
            "reconstruction_errors": {}, "compression_ratios": {}}
    bm = basis.get("masked")
    if bm is None:
        return diag
    first = next(iter(task_vectors.keys()))
    if param_name in task_vectors[first]:
        diag["original_shape"] = list(task_vectors[first][param_name].shape)
    if mask is not None:
        diag["masked_size"] = int(mask.sum().item())
        diag["unmasked_size"] = int((~mask).sum().item())
    else:
        diag["masked_size"] = np.prod(diag["original_shape"])
    diag["basis"] = {"k": bm["k"], "D": bm["D"], "N": bm["N"], "energy_retained": bm["energy_retained"]}
    errs = []
    for task in task_vectors.keys():
        if param_name not in task_vectors[task] or task not in compressed_params:
            continue
        delta = task_vectors[task][param_name]
        if mask is not None and mask.shape == delta.shape:
            x = apply_mask_to_tensor(delta, mask)
        else:
            x = delta.flatten()
        art = compressed_params[task]
        if art.get("masked") is None:
            continue
        c_high = art["masked"]["c_high_fp16"].to(dev).float()
        q = art["masked"]["c_low_quant"]
        c_low = quantizer.dequantize(q, device=dev).float()
        m = _fused_error(prepare_vector(x, dev), bm["U_high"], bm["U_low"], c_high, c_low, dev)
        errs.append(m["relative_error"])
        diag["reconstruction_errors"][task] = m
        diag["compression_ratios"][task] = estimate_compression_ratio(c_low, q)
    if errs:
        diag["mean_relative_error"] = float(np.mean(errs))
        diag["std_relative_error"] = float(np.std(errs))
        diag["max_relative_error"] = float(np.max(errs))
        diag["min_relative_error"] = float(np.min(errs))
    return diag


def _batched_errors(task_vectors, compressed_all, bases, masks) -> Dict[str, Dict]:
    """compute_parameter_diagnostics for every parameter whose artifacts still live in the buffers of a fused run: ONE
    pass over U and the N deltas per plan (svdq_diagnostics: all N error tuples of a parameter from a single read of
    its basis; masked parameters: svdq_diagnostics_masked, the selection made inside the pass) and one small D2H
    copy, instead of a mask compaction + payload upload + dequantize + fused-error launch per (parameter, task).
    Same dictionaries; parameters it cannot take (foreign artifacts, another mask or other tensors than the run
    compressed) are left to the per-parameter route."""
    from .merge import _batched_entry, _mask_identity
    plans = {}
    for name in bases.keys():
        if name not in compressed_all:
            continue
        got = _batched_entry(name, compressed_all, bases)
        if got is None:
            continue
        batch, i, meta = got
        mode = getattr(batch, "mode", None)
        if mode not in ("plain", "walk", "gather") or getattr(batch, "from_base", True):
            continue
        mask = masks.get(name)
        if mode == "plain":
            if mask is not None:
                continue
        elif (mask is None or batch.unit_start is None
              or getattr(batch, "mask_ident", {}).get(name) != _mask_identity(mask)
              or any(t in task_vectors and name in task_vectors[t] and mask.shape != task_vectors[t][name].shape
                     for t in batch.task_names[i])):
            # the selection is made with the mask the run compressed with (the SAME tensor: its bytes must not have been
            # edited in place since -- the unit starts date from the compression), and like the reference
            # (diagnostics.py:193-198) only when the mask has the delta's shape; anything else: per-parameter route
            continue
        # the error is measured against the tensors the CALLER passes: they must be the ones the plan still points at
        kept = batch.plan._keep[i] if batch.plan._keep is not None else None
        tasks_i = batch.task_names[i]
        if kept is None or any(t not in task_vectors or name not in task_vectors[t] or
                               task_vectors[t][name].data_ptr() != kept[j].data_ptr() or
                               task_vectors[t][name].numel() != kept[j].numel() for j, t in enumerate(tasks_i)):
            continue
        plans.setdefault(id(batch), (batch, []))[1].append((name, i, meta))
    out = {}
    for batch, items in plans.values():
        plan, small = batch.plan, batch.small
        with torch.cuda.device(plan.device):
            if batch.mode == "plain":
                res = plan.diagnostics(batch.table, batch.rows_dev)
            else:
                res = plan.diagnostics_masked(batch.table, batch.mask_table, batch.unit_start, batch.rows_dev)
            res = res.cpu().numpy()      # [P, N, 6]
        for name, i, meta in items:
            tasks_i = batch.task_names[i]
            pos = {t: j for j, t in enumerate(tasks_i)}
            first = next(iter(task_vectors.keys()))
            diag = {"param_name": name, "original_shape": None, "masked_size": 0, "unmasked_size": 0,
                    "reconstruction_errors": {}, "compression_ratios": {}}
            if name in task_vectors[first]:
                diag["original_shape"] = list(task_vectors[first][name].shape)
            k, r, rows = int(small.k[i]), int(small.r[i]), int(small.rows[i])
            if batch.mode == "plain":
                diag["masked_size"] = np.prod(diag["original_shape"])
            else:      # mask.sum() / (~mask).sum() (diagnostics.py:168-169): the counts the run kept on the device
                diag["masked_size"] = rows
                diag["unmasked_size"] = int(plan.rows[i]) - rows
            diag["basis"] = {"k": k, "D": rows, "N": plan.N, "energy_retained": float(small.energy[i])}
            nl, bits, stages = r - k, plan.bits_of(i), plan.S
            ratio = (nl * 4) / max(nl * bits / 8 * stages + 8 * stages, 1)      # estimate_compression_ratio, rtvq.py:142-161
            errs = []
            for task in task_vectors.keys():
                if name not in task_vectors[task] or task not in pos or task not in meta["have"]:
                    continue
                m = {key: float(v) for key, v in zip(_KEYS, res[i, pos[task]])}
                errs.append(m["relative_error"])
                diag["reconstruction_errors"][task] = m
                diag["compression_ratios"][task] = ratio
            if errs:
                diag["mean_relative_error"] = float(np.mean(errs))
                diag["std_relative_error"] = float(np.std(errs))
                diag["max_relative_error"] = float(np.max(errs))
                diag["min_relative_error"] = float(np.min(errs))
            out[name] = diag
    return out


def compute_all_diagnostics(task_vectors: Dict[str, Dict[str, torch.Tensor]], compressed_all: Dict[str, Dict],
                            bases: Dict[str, Dict], masks: Dict[str, torch.Tensor], config, device: str = "cpu") -> Dict:
    """Reference diagnostics.py:234-321."""
    quantizer = RTVQQuantizer(num_bits=config.svd_low_bits, num_stages=config.svd_rtvq_stages)
    out = {"config": {"svd_energy_threshold": config.svd_energy_threshold, "svd_max_rank": config.svd_max_rank,
                      "svd_low_bits": config.svd_low_bits, "svd_rtvq_stages": config.svd_rtvq_stages,
                      "svd_mask_strategy": config.svd_mask_strategy, "svd_weighting": config.svd_weighting},
           "per_parameter": {}, "summary": {}}
    pre = _batched_errors(task_vectors, compressed_all, bases, masks)
    for name in sorted(bases.keys()):
        if name not in compressed_all:
            continue
        if name in pre:
            out["per_parameter"][name] = pre[name]
            continue
        out["per_parameter"][name] = compute_parameter_diagnostics(name, task_vectors, compressed_all[name],
                                                                   bases[name], masks.get(name), quantizer, device)
    ranks, energies, mean_errs, ratios = [], [], [], []
    for d in out["per_parameter"].values():
        if "basis" in d:
            ranks.append(d["basis"]["k"])
            energies.append(d["basis"]["energy_retained"])
        if "mean_relative_error" in d:
            mean_errs.append(d["mean_relative_error"])
        if d.get("compression_ratios"):
            ratios.append(np.mean(list(d["compression_ratios"].values())))
    out["summary"] = {
        "num_parameters": len(out["per_parameter"]),
        "average_rank": float(np.mean(ranks)) if ranks else 0,
        "std_rank": float(np.std(ranks)) if ranks else 0,
        "average_energy_retained": float(np.mean(energies)) if energies else 0,
        "average_reconstruction_error": float(np.mean(mean_errs)) if mean_errs else 0,
        "average_compression_ratio": float(np.mean(ratios)) if ratios else 0,
    }
    return out


def compute_compression_statistics(task_vectors: Dict[str, Dict[str, torch.Tensor]], compressed_all: Dict[str, Dict],
                                   bases: Dict[str, Dict], config) -> Dict:
    """Reference diagnostics.py:385-566: byte accounting of the compressed representation (host arithmetic).
    Same keys; fp32 originals at 4 B/element, fp16 coefficients and bases at 2 B, RTVQ stages at
    ceil(n * bits / 8) + 8 B of scale/zero-point."""
    import math
    bits, stages = config.svd_low_bits, config.svd_rtvq_stages
    by_param = getattr(config, "svd_low_bits_by_param", None)   # mixed-width runs (extension, config #5)
    orig_task = {t: sum(d.numel() * 4 for d in tv.values()) for t, tv in task_vectors.items()}
    first = task_vectors[next(iter(task_vectors))]
    orig_param = {n: sum(tv[n].numel() * 4 for tv in task_vectors.values() if n in tv) for n in first.keys()}
    tot = {"fp16": 0, "rtvq": 0, "bases": 0}
    per_param, comp_param = {}, {}
    for name, basis in bases.items():
        if name not in compressed_all:
            continue
        ps = {"original_bytes": orig_param.get(name, 0), "compressed_bytes": 0, "fp16_high_energy_bytes": 0,
              "rtvq_low_energy_bytes": 0, "svd_bases_bytes": 0, "k": 0, "D": 0, "compression_ratio": 0}
        bm = basis.get("masked")
        if bm is not None:
            ps["k"], ps["D"] = bm.get("k", 0), bm.get("D", 0)
            ps["svd_bases_bytes"] = (bm["U_high"].numel() + bm["U_low"].numel()) * 2
            tot["bases"] += ps["svd_bases_bytes"]
            for art in compressed_all[name].values():
                if art is None or art.get("masked") is None:
                    continue
                ma = art["masked"]
                b16 = ma["c_high_fp16"].numel() * 2
                pb = int(by_param(name)) if by_param else bits
                bq = sum(math.ceil(p["quantized"].numel() * pb / 8) + 8 for p in ma["c_low_quant"].get("payloads", []))
                ps["fp16_high_energy_bytes"] += b16
                ps["rtvq_low_energy_bytes"] += bq
                tot["fp16"] += b16
                tot["rtvq"] += bq
        ps["compressed_bytes"] = ps["fp16_high_energy_bytes"] + ps["rtvq_low_energy_bytes"] + ps["svd_bases_bytes"]
        if ps["original_bytes"] > 0:
            ps["compression_ratio"] = ps["original_bytes"] / max(ps["compressed_bytes"], 1)
        per_param[name] = ps
        comp_param[name] = ps["compressed_bytes"]
    o_tot, c_tot = sum(orig_task.values()), tot["fp16"] + tot["rtvq"] + tot["bases"]
    return {
        "original": {"total_bytes": o_tot, "per_task_bytes": orig_task, "per_param_bytes": orig_param},
        "compressed": {"total_bytes": c_tot, "fp16_high_energy_bytes": tot["fp16"], "rtvq_low_energy_bytes": tot["rtvq"],
                       "svd_bases_bytes": tot["bases"], "per_task_bytes": {}, "per_param_bytes": comp_param},
        "per_parameter": per_param,
        "summary": {"original_size_mb": o_tot / (1024 * 1024), "compressed_size_mb": c_tot / (1024 * 1024),
                    "overall_compression_ratio": o_tot / max(c_tot, 1), "fp16_fraction": tot["fp16"] / max(c_tot, 1),
                    "rtvq_fraction": tot["rtvq"] / max(c_tot, 1), "bases_fraction": tot["bases"] / max(c_tot, 1),
                    "num_parameters": len(per_param), "num_tasks": len(task_vectors), "num_bits": bits,
                    "num_stages": stages}}


def print_detailed_compression_report(compression_stats: Dict, config=None, top_n_params: int = 5) -> None:
    """Short form of diagnostics.py:569-708: the numbers of the reference's report (overall sizes and shares, then
    the ``top_n_params`` largest parameters with their own breakdown), not its tutorial text and box drawing."""
    s = compression_stats["summary"]
    print(f"   compression: {s['original_size_mb']:.2f} MB -> {s['compressed_size_mb']:.2f} MB "
          f"({s['overall_compression_ratio']:.2f}x; bases {100 * s['bases_fraction']:.1f} %, fp16 "
          f"{100 * s['fp16_fraction']:.1f} %, RTVQ {100 * s['rtvq_fraction']:.1f} %)")
    per_param = compression_stats.get("per_parameter") or {}
    top = sorted(per_param.items(), key=lambda kv: kv[1].get("original_bytes", 0), reverse=True)[:max(int(top_n_params), 0)]
    bits = getattr(config, "svd_low_bits", None)
    for name, ps in top:
        shown = name[:50] + "..." if len(name) > 50 else name
        print(f"      {shown}: {ps.get('original_bytes', 0) / 1024:.2f} KB -> {ps.get('compressed_bytes', 0) / 1024:.2f} KB "
              f"({ps.get('compression_ratio', 0):.2f}x), k = {ps.get('k', 0)}, D = {ps.get('D', 0):,}; fp16 "
              f"{ps.get('fp16_high_energy_bytes', 0) / 1024:.2f} KB, "
              f"{str(bits) + '-bit ' if bits is not None else ''}RTVQ {ps.get('rtvq_low_energy_bytes', 0) / 1024:.2f} KB")


def print_diagnostics_summary(diagnostics: Dict) -> None:
    """Short form of diagnostics.py:711-."""
    s = diagnostics.get("summary", {})
    print(f"   reconstruction: average relative error {s.get('average_reconstruction_error', 0):.6f} over "
          f"{s.get('num_parameters', len(diagnostics.get('per_parameter', {})))} parameters")


def compute_coefficient_histograms(compressed_params: Dict[str, Dict], quantizer, num_bins: int = 50,
                                   device: str = "cpu") -> Dict:
    """Reference diagnostics.py:324-382: histograms of |c_high| and |dequantized c_low| over the tasks of one
    parameter (a few hundred scalars: numpy on the host after the GPU dequantization)."""
    import numpy as np
    highs, lows = [], []
    for art in compressed_params.values():
        if art is None or art.get("masked") is None:
            continue
        highs.append(art["masked"]["c_high_fp16"].float().flatten().cpu())
        lows.append(quantizer.dequantize(art["masked"]["c_low_quant"], device=device).float().flatten().cpu())
    if not highs:
        return {}
    out = {}
    for key, parts in (("c_high", highs), ("c_low", lows)):
        a = np.abs(torch.cat(parts).numpy())
        counts, edges = np.histogram(a, bins=num_bins)
        out[key] = {"counts": counts.tolist(), "bin_edges": edges.tolist(), "mean": float(np.mean(a)),
                    "std": float(np.std(a)), "max": float(np.max(a))}
    return out
