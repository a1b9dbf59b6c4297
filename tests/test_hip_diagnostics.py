"""
GPU tests of the diagnostics kernels that are FREE of the basis freedom: every number a
kernel reports for a (parameter, task) is compared with the reference's formula (diagnostics.py:72-117, 186-215)
evaluated in fp64 by the oracle on the very artifacts the kernel read -- the reference's own (tests/golden/diag.npz,
through svdq_recon_error) or the device's own (copied to the host: basis, fp16 c_high, codes / scale / zero_point
dequantized by the oracle's C restatement) -- all SIX keys, max_absolute_error included.

Tolerance (tests/helpers.py::diag_check): 2e-6 relative + the forward-error bound of the fp32 arithmetic in
``U_high.float() @ c_high + U_low.float() @ c_low`` itself, (r + 2) 2^-24 |U| |c| per element, which any fp32 evaluation
order is entitled to -- tests/test_oracle_golden.py::test_parameter_diagnostics_vs_reference_artifacts shows the
reference's own fp32 numbers use up to 7e-5 relative of it on max_absolute_error.  For centred runs (the
error is ~ ||mean||) the bound is ~3e-6 relative; spikes planted at block boundaries make a dropped or double-counted
row visible at 2e-6 whatever the bound.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, diag_check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sq():
    import svdq_amd
    return svdq_amd


def _host_artifacts(orc, plan, sm, p, t):
    """The device's own artifacts of (parameter p, task t) on the host, dequantized by the oracle."""
    k, r, rows = int(sm.k[p]), int(sm.r[p]), int(sm.rows[p])
    Uh, Ul, mean = plan.basis_tensors(p, k, r, rows)
    ch = torch.from_numpy(sm.c_high[p, t, :k].astype(np.float32))
    nl = r - k
    if nl > 0:
        cl = torch.from_numpy(orc.rtvq_dequantize({"codes": sm.codes[p, t, :, :nl], "scale": sm.scale[p, t],
                                                   "zero_point": sm.zero_point[p, t]}).reshape(-1).copy())
    else:
        cl = torch.zeros(0)
    return Uh.cpu(), Ul.cpu(), ch, cl, (mean.cpu() if mean is not None else None)


def test_recon_error_on_the_reference_artifacts(sq):
    """svdq_recon_error fed with the REFERENCE's basis and coefficients must return the reference's six numbers."""
    from oracle import svd_hybrid_oracle as orc
    from helpers import diag_fp32_bound
    from svdq_amd import diagnostics as dg
    dev = torch.device("cuda", 0)
    g = load_golden("diag.npz")
    for c in g["cases"]:
        for p in g[f"{c}__params"]:
            X = torch.from_numpy(g[f"{c}__x__{p}"])
            Uh, Ul = torch.from_numpy(g[f"{c}__U_high__{p}"]), torch.from_numpy(g[f"{c}__U_low__{p}"])
            ch, cl = torch.from_numpy(g[f"{c}__c_high_fp16__{p}"]).float(), torch.from_numpy(g[f"{c}__c_low_deq__{p}"])
            M = g[f"{c}__metrics__{p}"]
            for t in range(X.shape[0]):
                got = dg._fused_error(X[t].to(dev), Uh.to(dev), Ul.to(dev), ch[t], cl[t], dev)
                tol = diag_fp32_bound(Uh, Ul, ch[t], cl[t], X[t])
                for j, key in enumerate(orc.DIAG_KEYS):
                    assert abs(got[key] - M[t, j]) <= 2e-6 * abs(M[t, j]) + 2 * tol[key], (c, p, t, key, got[key], M[t, j])
                diag_check(got, X[t], Uh, Ul, ch[t], cl[t], what=(c, p, t))


def _spiked(vecs, sizes, gen):
    """Plant one large element per task at a block / unit boundary row (first, 255, 256, last, ...): its error term
    dominates max_absolute_error and shows in every sum, so a row that a kernel drops or counts twice cannot hide."""
    for p, D in enumerate(sizes):
        spots = [s for s in (0, 63, 64, 127, 128, 255, 256, 4095, 4096, 8191, 8192, D - 2, D - 1) if 0 <= s < D]
        for t, v in enumerate(vecs[p]):
            v[spots[(t + p) % len(spots)]] += 3.0 + 0.25 * t
    return vecs


@pytest.mark.parametrize("n_tasks,fp16,center", [(1, True, False), (2, False, True), (3, True, True), (4, True, False), (5, False, True), (8, True, True), (8, False, False), (12, True, False),
                                                 (16, False, True), (20, True, False), (20, False, True),
                                                 (28, True, True), (32, False, False)])
def test_plan_diagnostics_vs_fp64_oracle(sq, n_tasks, fp16, center):
    """svdq_diagnostics (k_diag, every task-count variant) and svdq_recon_error on the plan's own artifacts against
    the fp64 evaluation of the reference's formula; add_mean too (the extension, formula + mean)."""
    from oracle import svd_hybrid_oracle as orc
    from svdq_amd import diagnostics as dg
    from svdq_amd.pipeline import CompressPlan
    dev = torch.device("cuda", 0)
    sizes = [9000, 257, 4096 * 3 + 5, 1, 8192 + 300, 13, 40, 17]      # short last tiles after long parameters: stale LDS behind them
    gen = torch.Generator().manual_seed(n_tasks)
    vecs = [[d.clone() for d in orc.synthetic_deltas(D, n_tasks, 300 + i, rank=3)] for i, D in enumerate(sizes)]
    vecs = [[d.to(dev) for d in vs] for vs in _spiked(vecs, sizes, gen)]
    plan = CompressPlan(sizes, n_tasks, energy_threshold=0.9, max_rank=None, center=center, fp16=fp16, low_bits=4,
                        rtvq_stages=2, device=dev)
    table = plan.pointer_table(vecs)
    plan.run(table)
    sm = plan.fetch_small()
    res = plan.diagnostics(table).cpu().numpy()
    with_mean = plan.diagnostics(table, add_mean=True).cpu().numpy() if center else None
    for p, D in enumerate(sizes):
        for t in range(n_tasks):
            Uh, Ul, ch, cl, mean = _host_artifacts(orc, plan, sm, p, t)
            if not (torch.isfinite(cl).all() and np.isfinite(res[p, t]).all()):
                assert int(sm.r[p]) - int(sm.k[p]) <= 2 or D < n_tasks       # degenerate quantizer input
                continue
            x = vecs[p][t].cpu()
            diag_check(dict(zip(orc.DIAG_KEYS, res[p, t])), x, Uh, Ul, ch, cl, what=("k_diag", p, t))
            if t in (0, n_tasks - 1):
                got = dg._fused_error(vecs[p][t], Uh.to(dev), Ul.to(dev), ch, cl, dev)
                diag_check(got, x, Uh, Ul, ch, cl, what=("recon_error", p, t))
            if with_mean is not None:
                diag_check(dict(zip(orc.DIAG_KEYS, with_mean[p, t])), x, Uh, Ul, ch, cl, what=("add_mean", p, t), mean=mean)
    plan.close()


@pytest.mark.parametrize("n_tasks,fp16,inverted", [(2, True, True), (4, True, False), (8, True, True), (8, False, False), (16, True, False),
                                                   (20, True, False), (20, False, True), (32, True, False)])
def test_masked_plan_diagnostics_vs_fp64_oracle(sq, n_tasks, fp16, inverted):
    """svdq_diagnostics_masked (the walk form: apply_mask_to_tensor, mask_loader.py:651-679, inside the pass) against the
    fp64 formula on torch's own ``x[mask]``; signal polarity and the inverted one (noise regions take the cleared
    elements), dense and sparse masks, every task-count variant."""
    from oracle import svd_hybrid_oracle as orc
    from svdq_amd.mask_loader import MaskSet
    from svdq_amd.pipeline import CompressPlan
    dev = torch.device("cuda", 0)
    sizes = [9000, 300, 4096 * 3 + 5, 8192 + 300, 61, 150]
    dens = [0.9, 0.5, 0.97, 0.15, 0.3, 0.8]
    gen = torch.Generator().manual_seed(100 + n_tasks)
    vecs = [[d.clone() for d in orc.synthetic_deltas(D, n_tasks, 500 + i, rank=3)] for i, D in enumerate(sizes)]
    vecs = [[d.to(dev) for d in vs] for vs in _spiked(vecs, sizes, gen)]
    masks = [(torch.rand(D, generator=gen) < q) for D, q in zip(sizes, dens)]
    sel = [(~m if inverted else m) for m in masks]
    ms = MaskSet(sizes, dev)
    ct, cf = ms.count_scan([m.to(dev) for m in masks])
    comb = ms._s["mb"]                                       # the bool-byte tensors the scan read
    rows_dev = cf if inverted else ct
    plan = CompressPlan(sizes, n_tasks, energy_threshold=0.9, max_rank=None, center=False, fp16=fp16, low_bits=4,
                        rtvq_stages=2, device=dev)
    mtab = torch.tensor([c.data_ptr() for c in comb], dtype=torch.int64).to(dev)
    us = ms.unit_starts(plan, rows_dev, entry_map=[(q, inverted) for q in range(len(sizes))])
    table = plan.pointer_table(vecs)
    # compress the selected rows through compacted copies declared at the full size (the unit decomposition follows the
    # declared rows), so the test does not depend on which compress modes exist for this task count
    comp = [[torch.cat([v[s.to(dev)], torch.zeros(D - int(s.sum()), device=dev)]) for v in vs]
            for vs, s, D in zip(vecs, sel, sizes)]
    plan.run(plan.pointer_table(comp), rows_dev)
    sm = plan.fetch_small()
    assert [int(x) for x in sm.rows] == [int(s.sum()) for s in sel]
    res = plan.diagnostics_masked(table, mtab, us, rows_dev).cpu().numpy()
    for p, D in enumerate(sizes):
        for t in range(n_tasks):
            Uh, Ul, ch, cl, _ = _host_artifacts(orc, plan, sm, p, t)
            if not (torch.isfinite(cl).all() and np.isfinite(res[p, t]).all()):
                assert int(sm.r[p]) - int(sm.k[p]) <= 2
                continue
            x = vecs[p][t].cpu()[sel[p]]
            diag_check(dict(zip(orc.DIAG_KEYS, res[p, t])), x, Uh, Ul, ch, cl, what=("walk", p, t))
    plan.close()
