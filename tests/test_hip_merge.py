"""
GPU tests of the merge consumers: svdq_reconstruct / svdq_mask_expand and the
reference-shaped merge_parameter / merge_all_parameters / apply_merged_deltas, against vectors produced
by the reference (tests/golden/merge.npz) and against the oracle.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sq():
    import svdq_amd
    return svdq_amd


def test_reconstruct_kernel_vs_oracle(sq):
    from oracle import svd_hybrid_oracle as orc
    g = torch.Generator().manual_seed(1)
    for D, k, nl, fp16, with_mean in ((100, 2, 2, False, True), (70001, 3, 5, True, True), (4096, 8, 0, True, False),
                                      (1000003, 1, 7, True, True), (513, 0, 4, False, True)):
        Q, _ = torch.linalg.qr(torch.randn(D, max(k + nl, 1), generator=g))
        Uh, Ul = Q[:, :k].contiguous(), Q[:, k:k + nl].contiguous()
        if fp16:
            Uh, Ul = Uh.half(), Ul.half()
        ch, cl = torch.randn(k, generator=g), torch.randn(nl, generator=g)
        mean = torch.randn(D, 1, generator=g) if with_mean else None
        want = orc.reconstruct(ch, cl, Uh, Ul, mean).numpy()
        got = sq.reconstruct_from_coefficients(ch, cl, Uh.cuda(), Ul.cuda(), "cuda", mean=mean)
        assert got.device.type == "cuda" and got.shape == (D,)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_mask_expand_vs_torch(sq):
    g = torch.Generator().manual_seed(2)
    for shape, dens in (((40, 50), 0.6), ((2047,), 0.5), ((2049, 3), 0.03), ((1_000_003,), 0.94), ((4096,), 0.0),
                        ((4096,), 1.0), ((1,), 1.0)):
        mask = torch.rand(shape, generator=g) < dens
        full = torch.randn(shape, generator=g)
        sig, noi = full.flatten()[mask.flatten()], full.flatten()[~mask.flatten()]
        back = sq.reconstruct_from_masked(sig.cuda(), noi.cuda(), mask.cuda(), full.shape)
        assert torch.equal(back.cpu(), full)
        only = sq.reconstruct_from_masked(sig.cuda(), None, mask.cuda(), full.shape).cpu()
        want = torch.zeros_like(full)
        want[mask] = sig
        assert torch.equal(only, want)


def test_merge_vs_reference_vectors(sq):
    g = load_golden("merge.npz")
    tasks = [str(t) for t in g["tasks"]]
    params = [str(p) for p in g["params"]]
    weights = {t: float(w) for t, w in zip(tasks, g["weights"])}
    cfg = sq.SVDHybridConfig(svd_energy_threshold=0.9, svd_max_rank=2, svd_center=True, svd_fp16=True, svd_low_bits=4,
                             svd_rtvq_stages=2, svd_include_noise=True, svd_min_mask_size=10, svd_noise_shrink=0.5)
    tv = {t: {} for t in tasks}
    masks, shapes = {}, {}
    for p in params:
        shape = g[f"merged__{p}"].shape
        shapes[p] = torch.Size(shape)
        for i, t in enumerate(tasks):
            tv[t][p] = torch.from_numpy(g[f"in__{p}"][i]).view(*shape).cuda()
        if f"mask__{p}" in g:
            masks[p] = torch.from_numpy(g[f"mask__{p}"]).cuda()
    bases, comp = sq.run_basis_and_compress(tv, masks, cfg, "cuda")
    merged = sq.merge_all_parameters(comp, bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    assert sorted(merged.keys()) == sorted(params)
    for p in params:
        m = merged[p].cpu().numpy()
        assert m.shape == g[f"merged__{p}"].shape and np.isfinite(m).all()
        # vs the reference's merged delta: both carry independent quantization noise of the low coefficients
        assert float(np.mean((m - g[f"merged__{p}"]) ** 2)) <= 1e-6
        # and it approximates the exact weighted average as well as the reference does
        ex = g[f"exact__{p}"]
        err = np.linalg.norm(m - ex) / np.linalg.norm(ex)
        err_ref = np.linalg.norm(g[f"merged__{p}"] - ex) / np.linalg.norm(ex)
        assert err <= 1.5 * err_ref + 1e-3, (p, err, err_ref)
    base = {p: torch.from_numpy(g[f"base__{p}"]).cuda() for p in params}
    base["extra.buffer"] = torch.arange(5, dtype=torch.float32).cuda()
    final = sq.apply_merged_deltas(base, merged, device="cuda", verbose=False)
    assert set(final.keys()) == set(base.keys()) and torch.equal(final["extra.buffer"], base["extra.buffer"])
    assert final["extra.buffer"].data_ptr() != base["extra.buffer"].data_ptr()
    for p in params:
        assert torch.equal(final[p], base[p] + merged[p])
        assert float(np.mean((final[p].cpu().numpy() - g[f"final__{p}"]) ** 2)) <= 1e-6


def test_diagnostics_vs_reference_vectors(sq):
    """compute_all_diagnostics: same dict layout and (Q1 included) the same numbers as the reference."""
    import json
    g = load_golden("merge.npz")
    want = json.loads(str(g["diagnostics_json"]))
    tasks = [str(t) for t in g["tasks"]]
    params = [str(p) for p in g["params"]]
    cfg = sq.SVDHybridConfig(svd_energy_threshold=0.9, svd_max_rank=2, svd_center=True, svd_fp16=True, svd_low_bits=4,
                             svd_rtvq_stages=2, svd_include_noise=True, svd_min_mask_size=10, svd_noise_shrink=0.5)
    tv = {t: {} for t in tasks}
    masks = {}
    for p in params:
        shape = g[f"merged__{p}"].shape
        for i, t in enumerate(tasks):
            tv[t][p] = torch.from_numpy(g[f"in__{p}"][i]).view(*shape).cuda()
        if f"mask__{p}" in g:
            masks[p] = torch.from_numpy(g[f"mask__{p}"]).cuda()
    bases, comp = sq.run_basis_and_compress(tv, masks, cfg, "cuda")
    got = sq.compute_all_diagnostics(tv, comp, bases, masks, cfg, device="cuda")
    assert sorted(got.keys()) == sorted(want.keys()) and got["config"] == want["config"]
    assert sorted(got["summary"].keys()) == sorted(want["summary"].keys())
    for key in ("num_parameters", "average_rank", "std_rank"):
        assert got["summary"][key] == pytest.approx(want["summary"][key])
    assert got["summary"]["average_energy_retained"] == pytest.approx(want["summary"]["average_energy_retained"], abs=2e-5)
    # Q1: the mean is not added back, so the error is dominated by ||mean|| / ||x|| -- identical on both sides
    assert got["summary"]["average_reconstruction_error"] == pytest.approx(
        want["summary"]["average_reconstruction_error"], rel=2e-2)
    assert got["summary"]["average_compression_ratio"] == pytest.approx(want["summary"]["average_compression_ratio"])
    for p in params:
        gp, wp = got["per_parameter"][p], want["per_parameter"][p]
        assert sorted(gp.keys()) == sorted(wp.keys())
        assert gp["original_shape"] == wp["original_shape"] and int(gp["masked_size"]) == int(wp["masked_size"])
        assert int(gp["unmasked_size"]) == int(wp["unmasked_size"]) and gp["basis"]["k"] == wp["basis"]["k"]
        assert list(gp["reconstruction_errors"].keys()) == list(wp["reconstruction_errors"].keys()) == tasks
        for t in tasks:
            ge, we = gp["reconstruction_errors"][t], wp["reconstruction_errors"][t]
            assert sorted(ge.keys()) == sorted(we.keys())
            assert ge["original_norm"] == pytest.approx(we["original_norm"], rel=1e-6)
            for key in ("absolute_error", "relative_error", "mean_absolute_error", "reconstructed_norm"):
                assert ge[key] == pytest.approx(we[key], rel=5e-2), (p, t, key)
            assert gp["compression_ratios"][t] == pytest.approx(wp["compression_ratios"][t])
    # the two-vector form
    a = torch.from_numpy(g["in__a.weight"][0]).cuda()
    b = torch.from_numpy(g["merged__a.weight"]).flatten().cuda()
    em = sq.compute_reconstruction_error(a, b)
    want_em = g["err_metrics_T0_vs_merged"]
    keys = ("absolute_error", "relative_error", "max_absolute_error", "mean_absolute_error", "original_norm",
            "reconstructed_norm")
    for key, w in zip(keys, want_em):
        assert em[key] == pytest.approx(float(w), rel=1e-5), key
    z = sq.compute_reconstruction_error(torch.zeros(8), torch.ones(8))
    assert z["relative_error"] == 0 and z["original_norm"] == 0          # ||x|| <= 1e-10 -> 0 (diagnostics.py:107)


def test_artifacts_reload_end_to_end(sq, tmp_path):
    """Step 4-9 shape of the reference pipeline on the GPU: compress -> diagnostics -> save_all_artifacts ->
    reconstruct_from_artifacts (unmasked run, as the reference's reload supports) == in-memory merge."""
    from oracle import svd_hybrid_oracle as orc
    tasks = ["A", "B", "C", "D", "E", "F"]
    shapes = {"enc.w1": (64, 48), "enc.b1": (64,), "enc/w2": (32, 64)}
    tv = {t: {} for t in tasks}
    for pi, (n, shp) in enumerate(sorted(shapes.items())):
        for t, d in zip(tasks, orc.synthetic_deltas(int(np.prod(shp)), len(tasks), 500 + pi)):
            tv[t][n] = d.view(shp).cuda()
    cfg = sq.SVDHybridConfig(tasks=tasks, svd_energy_threshold=0.9, svd_max_rank=2, svd_low_bits=4, svd_rtvq_stages=2)
    bases, comp = sq.run_basis_and_compress(tv, None, cfg, "cuda")
    diag = sq.compute_all_diagnostics(tv, comp, bases, {}, cfg, device="cuda")
    weights = {t: 1.0 / len(tasks) for t in tasks}
    diag["task_weights"] = weights
    d = str(tmp_path / "art")
    sq.save_all_artifacts(bases, comp, diag, cfg, d)
    import os
    for fn in os.listdir(os.path.join(d, "coeffs")):      # payload tensors are views of one packed host buffer:
        assert os.path.getsize(os.path.join(d, "coeffs", fn)) < 32 * 1024, fn   # files hold only their own bytes
    base = {n: torch.randn(s).cuda() for n, s in shapes.items()}
    out_path = str(tmp_path / "merged.pt")
    res = sq.reconstruct_from_artifacts(d, base, out_path, device="cuda")
    merged = sq.merge_all_parameters(comp, bases, {}, weights, {n: torch.Size(s) for n, s in shapes.items()}, cfg,
                                     device="cuda", verbose=False)
    want = sq.apply_merged_deltas(base, merged, device="cuda", verbose=False)
    assert sorted(res["merged_state_dict"].keys()) == sorted(want.keys())
    for n in shapes:
        assert torch.equal(res["merged_state_dict"][n], want[n])
        exact = sum(weights[t] * tv[t][n] for t in tasks)
        rel = float((merged[n] - exact).norm() / exact.norm())
        assert rel < 0.05, (n, rel)
    saved = torch.load(out_path, weights_only=True)
    assert torch.equal(saved["enc.w1"].cuda(), want["enc.w1"])
    assert res["config"] == cfg and "per_parameter" in res["diagnostics"]
    # reload.py:60-139: a pre-saved merged model in the artifact directory wins; without one (and without a base
    # model path in the stored config) the reference's error is raised
    with pytest.raises(FileNotFoundError, match="Base model path not found"):
        sq.reload_merged_model_from_artifacts(d, device="cuda")
    torch.save({n: v.cpu() for n, v in want.items()}, os.path.join(d, "merged_state_dict.pt"))
    again = sq.reload_merged_model_from_artifacts(d, device="cpu")
    assert torch.equal(again["enc.w1"], want["enc.w1"].cpu())
    # diagnostics.py:324-382
    h = sq.compute_coefficient_histograms(comp["enc.w1"], sq.RTVQQuantizer(4, 2), num_bins=10, device="cuda")
    assert set(h) == {"c_high", "c_low"} and sum(h["c_high"]["counts"]) == len(tasks) * bases["enc.w1"]["masked"]["k"]
    assert len(h["c_low"]["bin_edges"]) == 11 and h["c_low"]["max"] >= h["c_low"]["mean"] >= 0


# ---------------------------------------------------------------------------- cluster weighting (config #5)
def _cluster_inputs(g):
    tasks = [str(t) for t in g["tasks"]]
    miss_t, miss_p = (str(x) for x in g["missing"])
    tv = {t: {} for t in tasks}
    for p in (str(x) for x in g["params"]):
        X = g[f"in__{p}"]
        for i, t in enumerate(tasks):
            if not (t == miss_t and p == miss_p):
                tv[t][p] = torch.from_numpy(X[i]).cuda()
    return tv, tasks


def test_task_gram_and_cluster_tasks_vs_reference(sq):
    from test_cluster_cpu import golden_gram, same_partition
    g = load_golden("cluster.npz")
    tv, tasks = _cluster_inputs(g)
    G, names = sq.task_gram(tv, "cuda")
    assert names == sorted(tasks)
    Gref, _ = golden_gram(g)
    # fp32 products summed in fp32 inside a 256-row block, fp64 across blocks
    np.testing.assert_allclose(G, Gref, rtol=2e-6, atol=2e-6 * np.abs(Gref).max())
    np.testing.assert_array_equal(G, G.T)
    for method in ("kmeans", "hierarchical"):
        for k in (2, 3):
            lab = sq.cluster_tasks(tv, k, method=method, device="cuda")
            assert same_partition([lab[t] for t in tasks], g[f"labels__{method}__k{k}"]), (method, k)
    assign = {t: int(l) for t, l in zip(tasks, g["labels__kmeans__k3"])}
    st = sq.compute_cluster_statistics(tv, assign, device="cuda")
    for i, cid in enumerate(g["stats__cids"]):
        for key in ("mean_distance_to_centroid", "max_distance_to_centroid", "min_distance_to_centroid"):
            assert st[int(cid)][key] == pytest.approx(float(g[f"stats__{key}"][i]), rel=5e-5, abs=1e-7)
    with pytest.raises(ValueError, match="Unknown clustering method"):
        sq.cluster_tasks(tv, 2, method="spectral")
    # task vectors averaged per cluster, then across clusters (clustering.py:319-425), on the GPU
    w = {t: float(x) for t, x in zip(tasks, g["w_cluster_perf"])}
    cperf = {i: float(v) for i, v in enumerate(g["cluster_perf"])}
    avg = sq.apply_weights_to_tensors({t: tv[t]["b.weight"] for t in tasks}, w)
    assert avg.is_cuda
    np.testing.assert_allclose(avg.cpu().numpy().reshape(-1), g["weighted_avg__b.weight"].reshape(-1), rtol=1e-5, atol=1e-8)
    final = sq.merge_cluster_results(sq.merge_by_cluster(tv, assign, w), cperf)
    for p in (str(x) for x in g["params"]):
        np.testing.assert_allclose(final[p].cpu().numpy().reshape(-1), g[f"mbc_final__{p}"].reshape(-1), rtol=1e-5,
                                   atol=1e-8)


def test_task_gram_full_size_linearity(sq):
    """Size-independent property at ViT-scale rows: Gram(a*X) = a^2 Gram(X), and Gram of a tensor split
    in two parameters equals the Gram of the whole."""
    torch.manual_seed(5)
    N, D = 8, 4096 * 1024 + 333
    X = [0.01 * torch.randn(D, device="cuda") for _ in range(N)]
    tv = {f"t{i}": {"w": X[i]} for i in range(N)}
    G, _ = sq.task_gram(tv, "cuda")
    cut = 1_000_000
    tv2 = {f"t{i}": {"a": X[i][:cut].clone(), "b": X[i][cut:].clone()} for i in range(N)}
    G2, _ = sq.task_gram(tv2, "cuda")
    tol = 1e-6 * np.abs(G).max()     # off-diagonal entries are ~1e-4 of the diagonal: absolute tolerance
    np.testing.assert_allclose(G2, G, rtol=1e-6, atol=tol)
    tv3 = {f"t{i}": {"w": 2.0 * X[i]} for i in range(N)}
    G3, _ = sq.task_gram(tv3, "cuda")
    np.testing.assert_array_equal(G3, 4.0 * G)           # scaling by 2 is exact in binary floating point
    ref = torch.stack(X).double()
    np.testing.assert_allclose(G, (ref @ ref.T).cpu().numpy(), rtol=2e-6, atol=tol)


def test_merge_with_clustering_vs_reference_vectors(sq):
    g = load_golden("merge.npz")
    gc = load_golden("cluster.npz")
    tasks = [str(t) for t in g["tasks"]]
    params = [str(p) for p in g["params"]]
    weights = {t: float(w) for t, w in zip(tasks, g["weights"])}
    cfg = sq.SVDHybridConfig(svd_energy_threshold=0.9, svd_max_rank=2, svd_center=True, svd_fp16=True, svd_low_bits=4,
                             svd_rtvq_stages=2, svd_include_noise=True, svd_min_mask_size=10, svd_noise_shrink=0.5)
    tv = {t: {} for t in tasks}
    masks, shapes = {}, {}
    for p in params:
        shape = g[f"merged__{p}"].shape
        shapes[p] = torch.Size(shape)
        for i, t in enumerate(tasks):
            tv[t][p] = torch.from_numpy(g[f"in__{p}"][i]).view(*shape).cuda()
        if f"mask__{p}" in g:
            masks[p] = torch.from_numpy(g[f"mask__{p}"]).cuda()
    bases, comp = sq.run_basis_and_compress(tv, masks, cfg, "cuda")
    assign = {t: int(c) for t, c in zip(tasks, gc["mwc__assign"])}
    merged = sq.merge_with_clustering(comp, bases, masks, weights, assign, shapes, cfg, device="cuda")
    assert sorted(merged.keys()) == sorted(params)
    for p in params:
        m = merged[p].cpu().numpy()
        assert m.shape == gc[f"mwc__{p}"].shape and np.isfinite(m).all()
        assert float(np.mean((m - gc[f"mwc__{p}"]) ** 2)) <= 1e-6
        ref = gc[f"mwc__{p}"]
        assert np.linalg.norm(m - ref) <= 0.25 * np.linalg.norm(ref), p   # same merge, independent 4-bit noise


# ---------------------------------------------------------------------------- batched consumers (svdq_merge.hip)
def _same_bits(a, b):
    """torch.equal that lets NaN equal NaN (F4: a 2-element c_low goes NaN in later stages, in every route alike)."""
    return a.shape == b.shape and torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(a.nan_to_num(), b.nan_to_num())


def _model_like(orc, tasks, with_masks):
    """A small model: matrices, vectors, ragged sizes; optionally two masked parameters (dense and sparse masks)."""
    shapes = {"blk.0.w": (96, 70), "blk.0.b": (96,), "blk.1.w": (300, 257), "emb": (1, 5000), "ln": (13,),
              "blk.1.b": (4097,)}
    tv = {t: {} for t in tasks}
    for pi, (n, shp) in enumerate(sorted(shapes.items())):
        for t, d in zip(tasks, orc.synthetic_deltas(int(np.prod(shp)), len(tasks), 900 + pi)):
            tv[t][n] = d.view(shp).cuda()
    masks = {}
    if with_masks:
        g = torch.Generator().manual_seed(4)
        masks["blk.1.w"] = (torch.rand(shapes["blk.1.w"], generator=g) < 0.9).cuda()
        masks["emb"] = (torch.rand(shapes["emb"], generator=g) < 0.2).cuda()
    return tv, masks, {n: torch.Size(s) for n, s in shapes.items()}


@pytest.mark.parametrize("with_masks,include_noise", [(False, False), (True, False), (True, True)])
def test_batched_merge_is_the_per_parameter_merge(sq, with_masks, include_noise):
    """merge_all_parameters / merge_with_clustering over the buffers of the fused run (two launches per plan: svdq_merge)
    give, bit for bit, what the per-parameter route gives from the materialised payload dictionaries; a task that
    lacks a parameter and non-uniform weights included."""
    from oracle import svd_hybrid_oracle as orc
    tasks = ["zeta", "alpha", "mid", "beta", "omega"]                       # insertion order != sorted order
    tv, masks, shapes = _model_like(orc, tasks, with_masks)
    del tv["mid"]["ln"]                                                     # one task lacks one parameter
    cfg = sq.SVDHybridConfig(tasks=tasks, svd_energy_threshold=0.9, svd_max_rank=2, svd_low_bits=4, svd_rtvq_stages=3,
                             svd_include_noise=include_noise, svd_noise_shrink=0.5, svd_min_mask_size=10)
    weights = {"zeta": 0.4, "alpha": 0.1, "mid": 0.2, "beta": 0.05, "omega": 0.25}
    bases, comp = sq.run_basis_and_compress(tv, masks, cfg, "cuda")
    plain = {n: {t: dict(a) if a is not None else None for t, a in v.items()} for n, v in comp.items()}   # no batch handle
    fast = sq.merge_all_parameters(comp, bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    from svdq_amd import merge as mg, mask_loader as mlo, pipeline as pipe
    calls = {"fused": 0, "expand": 0}
    orig_mm, orig_rm = pipe.CompressPlan.merge_masked, mlo.reconstruct_from_masked

    def spy_mm(self, *a, **kw):
        calls["fused"] += 1
        return orig_mm(self, *a, **kw)

    def spy_rm(*a, **kw):
        calls["expand"] += 1
        return orig_rm(*a, **kw)
    pipe.CompressPlan.merge_masked, mlo.reconstruct_from_masked = spy_mm, spy_rm
    try:
        assert len(mg._merge_batched(sorted(comp), comp, bases, masks, ([(weights, None)], None), shapes, cfg, "cuda")) \
            == len(comp)                                                     # every parameter took the batched route
    finally:
        pipe.CompressPlan.merge_masked, mlo.reconstruct_from_masked = orig_mm, orig_rm
    # masked parameters: scattered inside the streaming launch (signal plan, and the noise plan when there is one)
    assert calls == {"fused": (2 if include_noise else 1) if with_masks else 0, "expand": 0}
    slow = sq.merge_all_parameters(plain, bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    assert sorted(fast) == sorted(slow) == sorted(shapes)
    for n in shapes:
        assert fast[n].shape == shapes[n] and _same_bits(fast[n], slow[n]), n
        exact = sum(weights[t] * tv[t][n] for t in tasks if n in tv[t]) / sum(weights[t] for t in tasks if n in tv[t])
        if n not in masks and not torch.isnan(fast[n]).any():
            assert float((fast[n] - exact).norm() / exact.norm()) < 0.08, n
    on_cpu = sq.merge_all_parameters(comp, bases, masks, weights, shapes, cfg, verbose=False)     # default device="cpu"
    assert all(not v.is_cuda and _same_bits(v, fast[n].cpu()) for n, v in on_cpu.items())
    assign = {"zeta": 1, "alpha": 0, "mid": 1, "beta": 0, "omega": 1}
    cf = sq.merge_with_clustering(comp, bases, masks, weights, assign, shapes, cfg, device="cuda")
    assert mg._merge_with_clustering_batched(comp, bases, masks, weights, {1: ["zeta", "mid", "omega"], 0: ["alpha", "beta"]},
                                              shapes, cfg, "cuda") is not None
    cs = sq.merge_with_clustering(plain, bases, masks, weights, assign, shapes, cfg, device="cuda")
    for n in shapes:
        assert _same_bits(cf[n], cs[n]), n


def test_cluster_merge_when_a_whole_cluster_lacks_a_parameter(sq):
    """merge_with_clustering (reference merge.py:555-626) merges every parameter inside every cluster; a cluster none of
    whose members holds it contributes zeros and the cluster shares are normalised over ALL clusters
    (weighting.py:332-372) -- so such a parameter (a per-task head) comes out scaled by the shares of the clusters that
    do hold it, mean included.  The batched route must give the plain-dictionary route's bits (round-3 advice: it
    renormalised over the holding clusters instead)."""
    from oracle import svd_hybrid_oracle as orc
    tasks = ["zeta", "alpha", "mid", "beta", "omega", "psi", "chi"]
    assign = {"zeta": 1, "alpha": 0, "mid": 1, "beta": 0, "omega": 1, "psi": 1, "chi": 1}
    tv, masks, shapes = _model_like(orc, tasks, False)
    for t in ("alpha", "beta"):                                             # cluster 0 = {alpha, beta}: nobody has 'blk.0.b'
        del tv[t]["blk.0.b"]
    del tv["mid"]["emb"]                                                    # and one member of cluster 1 lacks 'emb'
    cfg = sq.SVDHybridConfig(tasks=tasks, svd_energy_threshold=0.9, svd_max_rank=2, svd_low_bits=4, svd_rtvq_stages=3,
                             svd_center=True)
    weights = {"zeta": 0.3, "alpha": 0.1, "mid": 0.2, "beta": 0.05, "omega": 0.15, "psi": 0.1, "chi": 0.1}
    bases, comp = sq.run_basis_and_compress(tv, None, cfg, "cuda")
    plain = {n: {t: dict(a) if a is not None else None for t, a in v.items()} for n, v in comp.items()}
    from svdq_amd import merge as mg
    members = {1: [t for t in tasks if assign[t] == 1], 0: [t for t in tasks if assign[t] == 0]}
    fast = mg._merge_with_clustering_batched(comp, bases, {}, weights, members, shapes, cfg, "cuda")
    assert fast is not None and sorted(fast) == sorted(shapes)
    slow = sq.merge_with_clustering(plain, bases, {}, weights, assign, shapes, cfg, device="cuda")
    for n in shapes:
        assert _same_bits(fast[n], slow[n]), n
    # 'blk.0.b' is held by cluster 1 only: the result is share_1 x (cluster 1's merge), not the merge itself
    only1 = sq.merge_all_parameters({n: {t: a for t, a in v.items() if assign[t] == 1} for n, v in plain.items()}, bases, {},
                                    {t: weights[t] for t in tasks if assign[t] == 1}, shapes, cfg, device="cuda", verbose=False)
    a, b = fast["blk.0.b"].double(), only1["blk.0.b"].double()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    ratio = float((a * b).sum() / (b * b).sum())
    perf = {c: np.mean([weights[t] for t in tasks if assign[t] == c]) for c in (0, 1)}
    share1 = float(np.exp(perf[1]) / (np.exp(perf[0]) + np.exp(perf[1])))
    assert ratio == pytest.approx(share1, rel=1e-5) and share1 < 0.6


def test_merged_tensors_are_the_callers_own(sq):
    """The reference returns fresh tensors: a second merge of the same artifacts (other weights) must not change what
    the first one returned (round-3 advice: the batched route handed out views of a buffer cached on the plan)."""
    from oracle import svd_hybrid_oracle as orc
    tasks = ["a", "b", "c", "d"]
    tv, masks, shapes = _model_like(orc, tasks, True)
    cfg = sq.SVDHybridConfig(tasks=tasks, svd_energy_threshold=0.9, svd_max_rank=2, svd_low_bits=4, svd_rtvq_stages=2)
    bases, comp = sq.run_basis_and_compress(tv, masks, cfg, "cuda")
    w1 = {"a": 0.7, "b": 0.1, "c": 0.1, "d": 0.1}
    w2 = {"a": 0.1, "b": 0.1, "c": 0.1, "d": 0.7}
    first = sq.merge_all_parameters(comp, bases, masks, w1, shapes, cfg, device="cuda", verbose=False)
    keep = {n: v.clone() for n, v in first.items()}
    second = sq.merge_all_parameters(comp, bases, masks, w2, shapes, cfg, device="cuda", verbose=False)
    assign = {"a": 0, "b": 0, "c": 1, "d": 1}
    third = sq.merge_with_clustering(comp, bases, masks, w1, assign, shapes, cfg, device="cuda")
    torch.cuda.synchronize()
    for n in shapes:
        assert _same_bits(first[n], keep[n]), n
        assert first[n].data_ptr() not in (second[n].data_ptr(), third[n].data_ptr())
        assert not torch.equal(first[n].nan_to_num(), second[n].nan_to_num()), n


def test_edited_dictionaries_leave_the_batched_route(sq):
    """The batched consumers answer from the buffers of the fused run; a caller who edits the dictionaries (drops a
    task, removes an artifact, switches the noise region off, swaps a basis entry) must get what the reference computes
    from the EDITED dictionaries (round-3 advice: the edits were ignored)."""
    from oracle import svd_hybrid_oracle as orc
    from svdq_amd import merge as mg
    tasks = ["a", "b", "c", "d", "e"]
    weights = {t: 0.2 for t in tasks}

    def fresh(include_noise=False):
        tv, masks, shapes = _model_like(orc, tasks, include_noise)
        cfg = sq.SVDHybridConfig(tasks=tasks, svd_energy_threshold=0.9, svd_max_rank=2, svd_low_bits=4, svd_rtvq_stages=2,
                                 svd_include_noise=include_noise, svd_min_mask_size=10)
        bases, comp = sq.run_basis_and_compress(tv, masks or None, cfg, "cuda")
        return tv, masks, shapes, cfg, bases, comp

    def plain_of(comp):
        return {n: {t: (dict(a) if a is not None else None) for t, a in v.items()} for n, v in comp.items()}

    # 1. a task deleted from one parameter
    tv, masks, shapes, cfg, bases, comp = fresh()
    assert mg._batched_entry("blk.0.w", comp, bases) is not None
    del comp["blk.0.w"]["c"]
    assert mg._batched_entry("blk.0.w", comp, bases) is None and mg._batched_entry("blk.0.b", comp, bases) is not None
    got = sq.merge_all_parameters(comp, bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    want = sq.merge_all_parameters(plain_of(comp), bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    assert all(_same_bits(got[n], want[n]) for n in shapes)
    full = sq.merge_all_parameters(plain_of(fresh()[5]), bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    assert not torch.equal(got["blk.0.w"], full["blk.0.w"]) and torch.equal(got["blk.0.b"], full["blk.0.b"])
    # 2. an artifact set to None inside a materialised entry (no LazyArtifacts method sees this edit)
    tv, masks, shapes, cfg, bases, comp = fresh()
    comp["blk.1.w"]["b"]["masked"] = None
    assert mg._batched_entry("blk.1.w", comp, bases) is None
    got = sq.merge_all_parameters(comp, bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    want = sq.merge_all_parameters(plain_of(comp), bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    assert all(_same_bits(got[n], want[n]) for n in shapes)
    # 3. the noise basis switched off for one parameter, a basis entry replaced for another
    tv, masks, shapes, cfg, bases, comp = fresh(include_noise=True)
    bases["blk.1.w"]["noise"] = None
    bases["emb"]["masked"]["U_high"] = bases["emb"]["masked"]["U_high"] * 2
    assert mg._batched_entry("blk.1.w", comp, bases) is None and mg._batched_entry("emb", comp, bases) is None
    got = sq.merge_all_parameters(comp, bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    want = sq.merge_all_parameters(plain_of(comp), bases, masks, weights, shapes, cfg, device="cuda", verbose=False)
    assert all(_same_bits(got[n], want[n]) for n in shapes)
    diag = sq.compute_all_diagnostics(tv, comp, bases, masks, cfg, device="cuda")      # and the diagnostics route follows
    assert set(diag["per_parameter"]) == set(shapes)


@pytest.mark.parametrize("n_tasks,fp16,inverted,n_sets", [(8, True, False, 1), (12, False, True, 2), (20, True, False, 2),
                                                          (20, False, True, 1), (32, True, False, 4)])
def test_plan_merge_masked_at_every_block_size(sq, n_tasks, fp16, inverted, n_sets):
    """svdq_merge_masked (k_merge_expand: 256-row chunks up to 16 tasks, 128-row chunks above -- round 3 stopped at 16) against
    merging in the compacted row space (svdq_merge) and torch's own boolean assignment: bit for bit, + base, both
    polarities, sparse and dense masks, ragged ends, several sets."""
    from oracle import svd_hybrid_oracle as orc
    from svdq_amd.mask_loader import MaskSet
    from svdq_amd.pipeline import CompressPlan
    dev = torch.device("cuda", 0)
    sizes = [9000, 300, 4096 * 3 + 5, 8192 + 300, 61]
    dens = [0.9, 0.5, 0.97, 0.15, 0.4]
    gen = torch.Generator().manual_seed(200 + n_tasks)
    vecs = [[d.to(dev) for d in orc.synthetic_deltas(D, n_tasks, 700 + i, rank=3)] for i, D in enumerate(sizes)]
    masks = [(torch.rand(D, generator=gen) < q) for D, q in zip(sizes, dens)]
    sel = [(~m if inverted else m).to(dev) for m in masks]
    ms = MaskSet(sizes, dev)
    ct, cf = ms.count_scan([m.to(dev) for m in masks])
    comb = ms._s["mb"]
    rows_dev = cf if inverted else ct
    plan = CompressPlan(sizes, n_tasks, energy_threshold=0.9, max_rank=None, center=True, fp16=fp16, low_bits=4,
                        rtvq_stages=2, device=dev)
    mtab = torch.tensor([c.data_ptr() for c in comb], dtype=torch.int64).to(dev)
    us = ms.unit_starts(plan, rows_dev, entry_map=[(q, inverted) for q in range(len(sizes))])
    comp = [[torch.cat([v[s_], torch.zeros(D - int(s_.sum()), device=dev)]) for v in vs] for vs, s_, D in zip(vecs, sel, sizes)]
    plan.run(plan.pointer_table(comp), rows_dev)
    sm = plan.fetch_small()
    w = torch.full((n_sets, n_tasks), -1.0)
    for t in range(n_tasks):
        w[t % n_sets, t] = 1.0 / len([u for u in range(n_tasks) if u % n_sets == t % n_sets])
    w = w.to(dev)
    share = torch.softmax(torch.arange(n_sets, dtype=torch.float32), 0).to(dev) if n_sets > 1 else None
    base = [torch.randn(D, generator=gen).to(dev) for D in sizes]
    btab = torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev)
    full = [torch.full((D,), float("nan"), device=dev) for D in sizes]
    otab = torch.tensor([f.data_ptr() for f in full], dtype=torch.int64).to(dev)
    plan.merge_masked(w, mtab, us, rows_dev, otab, set_share=share, fill=torch.ones(len(sizes), dtype=torch.int32, device=dev),
                      base_table=btab)
    cbuf, coffs, ctab = plan.new_merged_outputs()
    plan.merge(w, set_share=share, rows_dev=rows_dev, out_table=ctab)
    torch.cuda.synchronize()
    for p, D in enumerate(sizes):
        want = torch.zeros(D, device=dev)
        want[sel[p]] = cbuf[coffs[p]:coffs[p] + int(sm.rows[p])]
        assert _same_bits(full[p], base[p] + want), (p, D)
    plan.close()


@pytest.mark.parametrize("with_masks", [False, True])
def test_batched_diagnostics_match_per_parameter(sq, with_masks):
    """compute_all_diagnostics through svdq_diagnostics (one pass over U and the N deltas per plan; masked parameters:
    svdq_diagnostics_masked, the selection made inside the pass) against the per-(parameter, task) fused-error route:
    same dictionaries; the numbers of the two routes are two fp32 evaluation orders of the same formula (matrix pipe
    against fma chains) -- each is held to the fp64 oracle on its own artifacts in tests/test_hip_diagnostics.py, here they
    only have to agree to 1e-4."""
    from oracle import svd_hybrid_oracle as orc
    from svdq_amd import diagnostics as dg
    tasks = ["t3", "t1", "t2", "t0"]
    tv, masks, shapes = _model_like(orc, tasks, with_masks)
    tv = {t: {n: v.flatten() for n, v in d.items()} for t, d in tv.items()}      # resident flat tensors: the plan keeps them
    masks = {n: m.flatten() for n, m in masks.items()}
    cfg = sq.SVDHybridConfig(tasks=tasks, svd_energy_threshold=0.9, svd_max_rank=2, svd_low_bits=4, svd_rtvq_stages=2)
    bases, comp = sq.run_basis_and_compress(tv, masks or None, cfg, "cuda")
    assert len(dg._batched_errors(tv, comp, bases, masks)) == len(shapes)
    fast = sq.compute_all_diagnostics(tv, comp, bases, masks, cfg, device="cuda")
    plain = {n: {t: dict(a) for t, a in v.items()} for n, v in comp.items()}
    slow = sq.compute_all_diagnostics(tv, plain, bases, masks, cfg, device="cuda")
    assert list(fast["per_parameter"]) == list(slow["per_parameter"])
    for n in shapes:
        f, s = fast["per_parameter"][n], slow["per_parameter"][n]
        assert list(f.keys()) == list(s.keys()) and f["basis"] == s["basis"] and f["original_shape"] == s["original_shape"]
        assert int(f["masked_size"]) == int(s["masked_size"]) and f["compression_ratios"] == s["compression_ratios"]
        assert int(f["unmasked_size"]) == int(s["unmasked_size"])
        assert list(f["reconstruction_errors"]) == list(s["reconstruction_errors"]) == tasks
        for t in tasks:
            for key, v in s["reconstruction_errors"][t].items():
                assert f["reconstruction_errors"][t][key] == pytest.approx(v, rel=1e-4, abs=1e-12), (n, t, key)
        for key in ("mean_relative_error", "std_relative_error", "max_relative_error", "min_relative_error"):
            assert f[key] == pytest.approx(s[key], rel=1e-4, abs=1e-9)
    for key, v in slow["summary"].items():
        assert fast["summary"][key] == pytest.approx(v, rel=1e-4)
