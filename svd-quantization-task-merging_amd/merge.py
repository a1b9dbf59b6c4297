"""
Coefficient-space merge on the GPU with the reference's callables (SURVEY.md section 8 f1;
reference src/svd_hybrid/merge.py:61-552).  The heavy step -- U c + mean over every row of a
parameter -- is `svdq_reconstruct` (HBM-bound: reads the fp16 basis once, writes fp32); the masked
scatter is `svdq_mask_expand`.  Averaging N x N scalars per parameter is host-sized work on small
device tensors.  Weights come from weighting.py, cluster labels from clustering.py (Gram-based).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _native as nat
from .pipeline import prepare_vector, resolve_device, wants_cpu, _ptr, _stream_ptr
from .rtvq import RTVQQuantizer


def dequantize_and_average(compressed_coeffs: Dict[str, Dict], weights: Dict[str, float], quantizer: RTVQQuantizer,
                           region: str = "masked", device: str = "cpu"
                           ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Reference merge.py:61-141: tasks sorted by name, weights renormalised over present tasks."""
    names = sorted(compressed_coeffs.keys())
    highs, lows, ws = [], [], []
    for name in names:
        art = compressed_coeffs[name]
        if art is None or art.get(region) is None:
            continue
        ra = art[region]
        highs.append(ra["c_high_fp16"].to(device).float())
        lows.append(quantizer.dequantize(ra["c_low_quant"], device=device).float())
        ws.append(weights.get(name, 1.0 / len(names)))
    if not highs:
        return None, None
    tot = sum(ws)
    w = torch.tensor([x / tot for x in ws], device=device, dtype=torch.float32)
    # (stack * w).sum(0) task by task: rounded products added in task order -- the association the batched kernel
    # (k_merge_coeff) uses, so both routes give the same bits (torch's own reduction order depends on its launch shape)
    hi, lo = highs[0] * w[0], lows[0] * w[0]
    for j in range(1, len(highs)):
        hi = hi + highs[j] * w[j]
        lo = lo + lows[j] * w[j]
    return hi, lo


def reconstruct_from_coefficients(avg_c_high: torch.Tensor, avg_c_low: torch.Tensor, U_high: torch.Tensor,
                                  U_low: torch.Tensor, device: str = "cpu", mean: Optional[torch.Tensor] = None
                                  ) -> torch.Tensor:
    """Reference merge.py:144-194: U_high c_high + U_low c_low (+ mean); result on ``device`` (computed on the GPU)."""
    out = _reconstruct(avg_c_high, avg_c_low, U_high, U_low, mean, 1.0)
    return out.cpu() if wants_cpu(device) else out


def _reconstruct(avg_c_high, avg_c_low, U_high, U_low, mean, scale: float) -> torch.Tensor:
    """``scale`` folds the ``* noise_shrink`` of merge.py:284 into the same streaming pass."""
    lib = nat.lib()
    dev = resolve_device(U_high.device if U_high.is_cuda else "cuda")
    D = U_high.shape[0] if U_high.dim() == 2 else U_low.shape[0]
    k = U_high.shape[1] if U_high.dim() == 2 else 0
    nl = U_low.shape[1] if U_low.dim() == 2 else 0
    fp16 = (U_high.dtype == torch.float16) if k else (U_low.dtype == torch.float16)
    dt = torch.float16 if fp16 else torch.float32
    uh = U_high.to(device=dev, dtype=dt).contiguous() if k else None
    ul = U_low.to(device=dev, dtype=dt).contiguous() if nl else None
    coef = torch.cat([avg_c_high.to(dev).float().reshape(-1), avg_c_low.to(dev).float().reshape(-1)]).contiguous()
    if coef.numel() != k + nl:
        raise ValueError(f"Shape mismatch: {coef.numel()} coefficients for {k + nl} basis columns")
    m = prepare_vector(mean.squeeze() if mean.dim() > 1 else mean, dev) if mean is not None else None
    out = torch.empty(D, dtype=torch.float32, device=dev)
    if D == 0:
        return out
    with torch.cuda.device(dev):
        if k + nl <= 32:
            nat.check(lib.svdq_reconstruct(_ptr(uh), _ptr(ul), int(fp16), D, k, nl, _ptr(coef), _ptr(m), float(scale),
                                           _ptr(out), _stream_ptr()), "svdq_reconstruct")
        else:
            # wider than the path ever is (N <= 32): column groups, each adding onto the running sum through the
            # kernel's `mean` input; the scale goes on last
            groups = [(u[:, c0:c0 + 32].contiguous(), off + c0, min(32, n - c0))
                      for u, n, off in ((uh, k, 0), (ul, nl, k)) if n for c0 in range(0, n, 32)]
            acc, dst = m, (out, torch.empty_like(out))     # input and output of a launch never alias
            for i, (u, c0, cn) in enumerate(groups):
                last = i == len(groups) - 1
                nat.check(lib.svdq_reconstruct(_ptr(u), _ptr(None), int(fp16), D, cn, 0, _ptr(coef[c0:c0 + cn]),
                                               _ptr(acc), float(scale) if last else 1.0, _ptr(dst[i % 2]),
                                               _stream_ptr()), "svdq_reconstruct")
                acc = dst[i % 2]
            out = acc
    return out


def merge_parameter(param_name: str, compressed_params: Dict[str, Dict], basis: Dict, weights: Dict[str, float],
                    quantizer: RTVQQuantizer, original_shape: torch.Size, mask: Optional[torch.Tensor] = None,
                    include_noise: bool = False, noise_shrink: float = 1.0, device: str = "cpu") -> torch.Tensor:
    """Reference merge.py:197-301."""
    from .mask_loader import reconstruct_from_masked
    dev = resolve_device(device)
    merged_masked = merged_unmasked = None
    bm = basis.get("masked")
    if bm is not None:
        ch, cl = dequantize_and_average(compressed_params, weights, quantizer, region="masked", device=dev)
        if ch is not None:
            merged_masked = reconstruct_from_coefficients(ch, cl, bm["U_high"], bm["U_low"], device=dev,
                                                          mean=bm.get("mean"))
    if include_noise:
        bu = basis.get("noise")
        if bu is not None:
            ch, cl = dequantize_and_average(compressed_params, weights, quantizer, region="unmasked", device=dev)
            if ch is not None:
                merged_unmasked = _reconstruct(ch, cl, bu["U_high"], bu["U_low"], bu.get("mean"), noise_shrink)
    if mask is not None and merged_masked is not None:
        res = reconstruct_from_masked(merged_masked, merged_unmasked, mask, original_shape)
    elif merged_masked is not None:
        res = merged_masked.view(original_shape)
    else:
        res = torch.zeros(original_shape, device=dev)
    return res.cpu() if wants_cpu(device) else res      # the reference returns on `device` (default "cpu")


def _batched_entry(name, compressed_all, bases):
    """(plan batch, index, meta) when both dictionaries of ``name`` still are what the fused run handed out: the
    LazyArtifacts themselves drop their handle on any edit through them; what an edit can reach WITHOUT going through
    them is checked here -- the noise entry of the (plain) per-parameter basis dictionary, and, once a compressed entry
    has been materialised, the per-task dictionaries it handed out (a deleted task, an artifact replaced or set to
    None).  In-place edits of payload tensors are not detectable; the reference has no such use."""
    from .driver import LazyArtifacts
    from .pipeline import task_artifact
    b, ca = bases.get(name), compressed_all.get(name)
    bm = b.get("masked") if isinstance(b, dict) else None
    if not (isinstance(bm, LazyArtifacts) and bm._batch is not None and isinstance(ca, LazyArtifacts)
            and ca._meta is not None and ca._batch is not None and ca._batch[0] is bm._batch[0]
            and ca._batch[1] == bm._batch[1]):
        return None
    meta = ca._meta
    bn = b.get("noise")
    if meta["noise"] is not None:
        if not (isinstance(bn, LazyArtifacts) and bn._batch is not None and bn._batch[0] is meta["noise"][0]
                and bn._batch[1] == meta["noise"][1]):
            return None
    elif bn is not None:
        return None
    if ca._fill is None:      # materialised: somebody may hold (and have edited) the per-task dictionaries
        batch, i = bm._batch
        if list(dict.keys(ca)) != list(meta["have"]):
            return None
        pos = {t: j for j, t in enumerate(batch.task_names[i])}
        npos = None
        if meta["noise"] is not None:
            nb, j = meta["noise"]
            npos = {t: q for q, t in enumerate(nb.task_names[j])}
        for t in meta["have"]:
            art = dict.__getitem__(ca, t)
            if not isinstance(art, dict) or set(art.keys()) != {"masked", "unmasked"}:
                return None
            want_m = task_artifact(batch.plan, batch.small, i, pos[t]) if t in pos else None
            want_u = task_artifact(nb.plan, nb.small, j, npos[t]) if (npos is not None and t in npos) else None
            if art["masked"] is not want_m or art["unmasked"] is not want_u:
                return None
    return bm._batch[0], bm._batch[1], meta


def _task_weights(have, tasks_i, members, weights):
    """One row of the weight table of k_merge_coeff: the reference's renormalised weights (merge.py:89-124) of the
    tasks of ``members`` (None = all) that have the parameter, in the plan's task order; -1 = not in the set."""
    import numpy as np
    names = sorted(t for t in have if members is None or t in members)
    present = [t for t in names if t in tasks_i]
    row = np.full(len(tasks_i), -1.0, dtype=np.float32)
    if not present:
        return row, False
    ws = [weights.get(t, 1.0 / len(names)) for t in present]
    tot = sum(ws)
    pos = {t: j for j, t in enumerate(tasks_i)}
    for t, x in zip(present, ws):
        row[pos[t]] = np.float32(x / tot)
    return row, True


def _mask_identity(mask) -> Optional[tuple]:
    return None if mask is None else (mask.data_ptr(), mask.numel(), str(mask.device), mask.dtype)


def _merge_batched(names, compressed_all, bases, masks, sets, original_shapes, config, device):
    """merge_all_parameters / merge_with_clustering for the parameters whose artifacts still live in the buffers of a
    fused run: per plan ONE coefficient kernel + ONE streaming reconstruction (svdq_merge) instead of, per parameter
    and task, a host->device copy of the payloads, a dequantize launch, a stack/sum and a reconstruct launch.  Masked
    parameters: the same two launches per plan with reconstruct_from_masked inside the streaming one
    (svdq_merge_masked: signal and noise regions write their own rows of the full tensor).
    ``sets``: [(weights dict, members or None)] and, for more than one set, their shares (device tensor).
    Returns {name: merged delta} for the names it could take; the rest goes the per-parameter way."""
    import numpy as np
    set_list, shares = sets
    S = len(set_list)
    if S > 8:
        return {}
    dev = resolve_device(device)
    jobs = {}      # id(plan) -> (plan batch, {entry index: (name, region, meta)})
    for name in names:
        got = _batched_entry(name, compressed_all, bases)
        if got is None:
            continue
        batch, i, meta = got
        jobs.setdefault(id(batch.plan), (batch, {}))[1][i] = (name, "masked", meta)
        if config.svd_include_noise and meta["noise"] is not None:
            nb, j = meta["noise"]
            jobs.setdefault(id(nb.plan), (nb, {}))[1][j] = (name, "noise", meta)
    # the per-plan tables, and which (parameter, region) pairs have something to merge
    tables, live = {}, {}
    for key, (batch, entries) in jobs.items():
        plan, small = batch.plan, batch.small
        P, N = plan.P, plan.N
        wt = np.full((P, S, N), -1.0, dtype=np.float32)
        order = np.tile(np.arange(N, dtype=np.int32), (P, 1))
        scale = np.ones(P, dtype=np.float32)
        share_ok = np.zeros((P, S), dtype=bool)
        for i, (name, region, meta) in entries.items():
            tasks_i = batch.task_names[i]
            order[i] = np.argsort(np.array(tasks_i, dtype=object), kind="stable").astype(np.int32)
            for s_, (w, members) in enumerate(set_list):
                wt[i, s_], share_ok[i, s_] = _task_weights(meta["have"], tasks_i, members, w)
            if region == "noise":
                scale[i] = np.float32(config.svd_noise_shrink)
            if int(small.rows[i]) > 0 and share_ok[i].any():
                live.setdefault(name, {})[region] = (key, i)
        tables[key] = (wt, order, scale, share_ok)
    # masked parameters go through the fused scatter when the caller's mask is the one the run compressed with
    full, fused = {}, set()
    for name, regs in live.items():
        if "masked" not in regs:
            continue
        batch = jobs[regs["masked"][0]][0]
        if getattr(batch, "mode", "plain") == "plain":
            continue
        ident = getattr(batch, "mask_ident", {}).get(name)
        nb = jobs[regs["noise"][0]][0] if "noise" in regs else None
        if (ident is None or ident != _mask_identity(masks.get(name)) or batch.unit_start is None
                or (nb is not None and (nb.unit_start is None or getattr(nb, "mask_ident", {}).get(name) != ident))):
            continue
        fused.add(name)
    pieces = {}    # name -> {"masked": tensor, "noise": tensor} (compacted rows; the unfused way)
    with torch.cuda.device(dev):
        for key, (batch, entries) in jobs.items():
            plan, small = batch.plan, batch.small
            P = plan.P
            wt, order, scale, share_ok = tables[key]
            wt_d = torch.from_numpy(wt).to(plan.device)
            ord_d = torch.from_numpy(order).to(plan.device)
            sc_d = torch.from_numpy(scale).to(plan.device)
            sh_d = None
            if S > 1:
                # merge_with_clustering (merge.py:555-626) merges EVERY parameter inside every cluster -- a cluster none of
                # whose members has it contributes merge_parameter's zeros (merge.py:297-299) -- so
                # apply_weights_to_tensors (weighting.py:332-372) normalises the shares over ALL clusters, and a
                # parameter only some clusters hold comes out scaled by the sum of THEIR shares (its mean included):
                # shares / shares.sum() for the sets that have it, "skip" (-1) for the others -- no renormalisation
                ok = torch.from_numpy(share_ok).to(plan.device)
                sh = shares.to(plan.device).view(S)
                tot = torch.zeros((), dtype=torch.float32, device=plan.device)
                for s_ in range(S):      # w.sum() of <= 8 values, in set order
                    tot = tot + sh[s_]
                sh_d = torch.where(ok, (sh / tot).view(1, S).expand(P, S),
                                   torch.full((), -1.0, device=plan.device)).contiguous()
            rows_dev = plan.small[plan.layout.rows_off:plan.layout.rows_off + 8 * P].view(torch.int64)
            mine = {i: e for i, e in entries.items() if live.get(e[0], {}).get(e[1]) == (key, i)}
            take = {i: e for i, e in mine.items() if e[0] in fused}
            if take:
                out_tab = np.zeros(P, dtype=np.int64)
                fill = np.zeros(P, dtype=np.int32)
                for i, (name, region, meta) in take.items():
                    if name not in full:
                        full[name] = torch.empty(plan.rows[i], dtype=torch.float32, device=plan.device)
                    out_tab[i] = full[name].data_ptr()
                    fill[i] = int(region == "masked" and "noise" not in live[name])
                plan.merge_masked(wt_d, batch.mask_table, batch.unit_start, rows_dev,
                                  torch.from_numpy(out_tab).to(plan.device), order=ord_d, set_share=sh_d, scale=sc_d,
                                  fill=torch.from_numpy(fill).to(plan.device))
            if len(take) < len(mine):
                # a buffer of this call's own: the reference returns fresh tensors, and a later merge of the same plan
                # (other weights, an ablation) must not overwrite what this one handed out
                buf, offs, otab = plan.new_merged_outputs()
                plan.merge(wt_d, order=ord_d, set_share=sh_d, scale=sc_d, rows_dev=rows_dev, out_table=otab)
                for i, (name, region, meta) in mine.items():
                    if i not in take:
                        pieces.setdefault(name, {})[region] = buf[offs[i]:offs[i] + int(small.rows[i])]
    from .mask_loader import reconstruct_from_masked
    out = {name: t.view(original_shapes[name]) for name, t in full.items()}
    for name, pc in pieces.items():
        if "masked" not in pc:
            continue
        mask = masks.get(name)
        if mask is not None:
            res = reconstruct_from_masked(pc["masked"], pc.get("noise"), mask, original_shapes[name])
        else:
            res = pc["masked"].view(original_shapes[name])
        out[name] = res
    return out


def merge_all_parameters(compressed_all: Dict[str, Dict[str, Dict]], bases: Dict[str, Dict],
                         masks: Dict[str, torch.Tensor], weights: Dict[str, float],
                         original_shapes: Dict[str, torch.Size], config, device: str = "cpu",
                         verbose: bool = True) -> Dict[str, torch.Tensor]:
    """Reference merge.py:304-426: parameters in sorted order, one quantizer for the run."""
    quantizer = RTVQQuantizer(num_bits=config.svd_low_bits, num_stages=config.svd_rtvq_stages)
    names = sorted(compressed_all.keys())
    # parameters whose artifacts still live in a fused run's buffers: two launches per plan (svdq_merge)
    fast = _merge_batched(names, compressed_all, bases, masks, ([(weights, None)], None), original_shapes, config, device)
    merged = {}
    for name in names:
        if name in fast:
            merged[name] = fast[name].cpu() if wants_cpu(device) else fast[name]
            continue
        merged[name] = merge_parameter(name, compressed_all[name], bases[name], weights, quantizer,
                                       original_shapes[name], mask=masks.get(name),
                                       include_noise=config.svd_include_noise, noise_shrink=config.svd_noise_shrink,
                                       device=device)
    if verbose:
        print(f"   merged {len(merged)} parameters")
    return merged


def apply_merged_deltas(base_state_dict: Dict[str, torch.Tensor], merged_deltas: Dict[str, torch.Tensor],
                        device: str = "cpu", verbose: bool = True) -> Dict[str, torch.Tensor]:
    """Reference merge.py:429-552: merged[param] = base[param] + delta[param]; others are cloned."""
    out = {}
    for name, base in base_state_dict.items():
        if name in merged_deltas:
            out[name] = base + merged_deltas[name].to(base.device)
        else:
            out[name] = base.clone()
    if verbose:
        print(f"   applied {sum(n in merged_deltas for n in base_state_dict)} merged deltas")
    return out


def _merge_with_clustering_batched(compressed_all, bases, masks, weights, clusters, original_shapes, config, device):
    """merge_with_clustering when EVERY parameter can take the batched route: the per-cluster merges and the share-weighted
    sum over clusters happen inside one streaming pass per plan (the merge is linear in the coefficients).  None
    otherwise (then the per-cluster loop below runs)."""
    names = sorted(compressed_all.keys())
    if not names or len(clusters) > 8 or any(_batched_entry(n, compressed_all, bases) is None for n in names):
        return None
    cids = sorted(clusters.keys())      # apply_weights_to_tensors adds the clusters in sorted-key order
    sets, perf = [], []
    for cid in cids:
        members = clusters[cid]
        w = {m: weights.get(m, 1.0) for m in members}
        total = sum(w.values())
        sets.append(({m: v / total for m, v in w.items()}, set(members)))
        perf.append(sum(weights.get(m, 1.0) for m in members) / len(members))
    # clustering.merge_cluster_results: softmax over the clusters in their first-seen order, looked up per cluster id
    seen = list(clusters.keys())
    sm = torch.softmax(torch.tensor([perf[cids.index(c)] for c in seen]), dim=0)
    by_cid = {c: sm[j].item() for j, c in enumerate(seen)}
    dev = resolve_device(device)
    shares = torch.tensor([by_cid[c] for c in cids], device=dev, dtype=torch.float32)
    fast = _merge_batched(names, compressed_all, bases, masks, (sets, shares), original_shapes, config, device)
    if len(fast) != len(names):
        return None
    return {n: (t.cpu() if wants_cpu(device) else t) for n, t in fast.items()}


def merge_with_clustering(compressed_all: Dict[str, Dict[str, Dict]], bases: Dict[str, Dict],
                          masks: Dict[str, torch.Tensor], weights: Dict[str, float],
                          cluster_assignments: Dict[str, int], original_shapes: Dict[str, torch.Size], config,
                          device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Reference merge.py:555-626: merge inside each cluster (weights renormalised over its members), then
    average the cluster results with softmax(mean member weight) shares (clustering.py:374-425)."""
    from .clustering import get_cluster_members, merge_cluster_results
    clusters = get_cluster_members(cluster_assignments)
    fast = _merge_with_clustering_batched(compressed_all, bases, masks, weights, clusters, original_shapes, config, device)
    if fast is not None:
        return fast
    per_cluster, performance = {}, {}
    for cid, members in clusters.items():
        w = {m: weights.get(m, 1.0) for m in members}
        total = sum(w.values())
        w = {m: v / total for m, v in w.items()}
        subset = {p: {m: arts[m] for m in members if m in arts} for p, arts in compressed_all.items()}
        per_cluster[cid] = merge_all_parameters(subset, bases, masks, w, original_shapes, config, device,
                                                verbose=False)
        performance[cid] = sum(weights.get(m, 1.0) for m in members) / len(members)
    return merge_cluster_results(per_cluster, performance, device)
