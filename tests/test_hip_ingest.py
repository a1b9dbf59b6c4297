"""
GPU tests of the ingest / whole-tensor-quantization front end: svdq_ingest,
svdq_tvq_quantize, svdq_tvq_dequantize behind the reference-shaped TaskVector / Quantized* classes,
against vectors produced by the reference (tests/golden/tvq.npz), against the oracle at 4 M elements and
through size-independent properties at ViT-L sizes.  Integer codes, scales and fp32 results: bit-exact.
"""
import numpy as np
import pytest
import torch

from helpers import bits_equal, load_golden
from oracle import svd_hybrid_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sq():
    import svdq_amd
    return svdq_amd


@pytest.fixture(scope="module")
def g():
    return load_golden("tvq.npz")


def _states(g):
    keys = [str(k) for k in g["keys"]]
    base = {k: torch.from_numpy(g[f"base__{k}"]) for k in keys}
    fts = {}
    for t in (str(x) for x in g["tasks"]):
        fts[t] = {k: torch.from_numpy(g[f"ft__{t}__{k}"]) for k in keys if f"ft__{t}__{k}" in g}
    return base, fts


def test_task_vectors_vs_reference(sq, g):
    base, fts = _states(g)
    many = sq.TaskVector.from_many(base, fts)                     # one launch group for the 3 tasks
    for t, ft in fts.items():
        tv = sq.TaskVector(base, ft, task_name=t, verbose=False)
        ref_keys = [str(k) for k in g[f"tv_keys__{t}"]]
        assert list(tv.vector.keys()) == ref_keys == list(many[t].vector.keys())
        for k in ref_keys:
            assert tv.vector[k].is_cuda and tv.vector[k].shape == base[k].shape
            assert bits_equal(tv.vector[k].cpu().numpy(), g[f"tv__{t}__{k}"]), (t, k)
            assert torch.equal(tv.vector[k], many[t].vector[k])
        ctv = sq.compute_task_vector(base, ft, device="cuda")
        assert list(ctv.keys()) == [str(k) for k in g[f"ctv_keys__{t}"]]     # no dtype skipping in the loader
        for k in ref_keys:
            assert torch.equal(ctv[k], tv.vector[k])
    a = sq.TaskVector(base, fts["A"], task_name="A", verbose=False)
    b = sq.TaskVector(base, fts["B"], task_name="B", verbose=False)
    s = (a + b) * 0.5
    assert s.task_name == str(g["sum_name"]) and list(s.vector.keys()) == [str(k) for k in g["sum_keys"]]
    for k in s.vector:
        assert bits_equal(s.vector[k].cpu().numpy(), g[f"sum__{k}"])
    assert (0.5 * (a - a)).vector["w1"].abs().max().item() == 0.0
    applied = a.apply_to(base, verbose=False)
    assert list(applied.keys()) == list(base.keys())
    for k, v in applied.items():
        assert np.array_equal(v.cpu().numpy(), g[f"applied__{k}"]), k
    assert sq.get_parameter_names({t: many[t].vector for t in many}) == sorted(g["tv_keys__A"].tolist())
    by_param = sq.organize_by_parameter({t: many[t].vector for t in many})
    assert sorted(by_param["b1"].keys()) == ["A", "C"] and sorted(by_param["w2"].keys()) == ["A", "B"]
    deltas, names = sq.flatten_task_deltas({t: many[t].vector for t in many}, "w2")
    assert names == ["A", "B"] and deltas[0].dim() == 1


def _check_payloads(pay, g, prefix):
    assert list(pay.keys()) == [str(k) for k in g[f"{prefix}__keys"]]
    for k, p in pay.items():
        ref_q = g[f"{prefix}__q__{k}"]
        q = p["quantized"].cpu().numpy()
        assert q.dtype == ref_q.dtype and q.shape == ref_q.shape and np.array_equal(q, ref_q), (prefix, k)
        assert p["scale"].dim() == 0 and bits_equal(p["scale"].cpu().numpy(), g[f"{prefix}__scale__{k}"]), (prefix, k)
        if f"{prefix}__zp__{k}" in g:
            assert bits_equal(p["zero_point"].cpu().numpy(), g[f"{prefix}__zp__{k}"]), (prefix, k)
        else:
            assert "zero_point" not in p
        assert tuple(p["shape"]) == ref_q.shape


@pytest.mark.parametrize("method", ["asymmetric", "absmax"])
def test_quantized_family_vs_reference(sq, g, method):
    base, fts = _states(g)
    for qbit in (8, 4, 3):
        tag = f"{method}{qbit}"
        qf = sq.QuantizedFinetunedModel(fts["A"], qbit=qbit, method=method)
        _check_payloads(qf.quantized_weights, g, f"qf__{tag}")
        deq = qf.dequantize()
        for k, v in deq.items():
            assert bits_equal(v.cpu().numpy(), g[f"qf__{tag}__deq__{k}"]), (tag, k)
        tv = qf.get_task_vector(base)
        assert list(tv.keys()) == list(deq.keys())
        for k, v in tv.items():
            assert bits_equal(v.cpu().numpy(), g[f"qf__{tag}__tv__{k}"]), (tag, k)
    tvA = sq.TaskVector(base, fts["A"], task_name="A", verbose=False)
    qb = sq.QuantizedBaseAndTaskVector(base, tvA, base_qbit=8, task_qbit=4, method=method)
    _check_payloads(qb.quantized_base, g, f"qb__{method}__base")
    _check_payloads(qb.quantized_task, g, f"qb__{method}__task")
    for k, v in qb.dequantize().items():
        assert bits_equal(v.cpu().numpy(), g[f"qb__{method}__deq__{k}"]), k
    qt = sq.QuantizedTaskVector(qb.quantized_task, method=method)
    for k, v in qt.dequantize().items():
        assert bits_equal(v.cpu().numpy(), g[f"qt__{method}__deq__{k}"]), k
    applied = qt.apply_to(base)
    assert list(applied.keys()) == list(base.keys())
    for k, v in applied.items():
        ref = g[f"qt__{method}__applied__{k}"]
        assert np.array_equal(v.cpu().numpy(), ref) if ref.dtype.kind in "iu" else bits_equal(v.cpu().numpy(), ref), k
    qt2 = sq.QuantizedTaskVector.from_task_vector(tvA, qbit=4, method=method)
    for k, p in qt2.quantized_deltas.items():
        assert torch.equal(p["quantized"], qb.quantized_task[k]["quantized"])


def test_quantizer_rejects_unsupported_widths(sq):
    with pytest.raises(NotImplementedError):
        sq.quantize_state_dict({"w": torch.randn(8)}, qbit=16)
    with pytest.raises(NotImplementedError):
        sq.quantize_state_dict({"w": torch.randn(8)}, qbit=1, method="absmax")


def test_tvq_vs_oracle_4m_elements(sq):
    """the standalone large-tensor form, n = 4 194 304 (+3 for the scalar tail)."""
    torch.manual_seed(11)
    x = 0.02 * torch.randn(4 * 1024 * 1024 + 3)
    x[12345] = 0.31
    for method, bits in (("asymmetric", 8), ("asymmetric", 4), ("asymmetric", 2), ("absmax", 8), ("absmax", 4)):
        pay = sq.quantize_state_dict({"x": x}, qbit=bits, method=method)["x"]
        if method == "asymmetric":
            q, sc, zp = orc.asym_quantize(x.numpy(), bits)
            assert bits_equal(pay["zero_point"].cpu().numpy(), np.float32(zp))
            deq_ref = orc.asym_dequantize(q, sc, zp)
        else:
            q, sc = orc.absmax_quantize(x.numpy(), bits)
            deq_ref = orc.absmax_dequantize(q, sc)
        assert np.array_equal(pay["quantized"].cpu().numpy(), q), (method, bits)
        assert bits_equal(pay["scale"].cpu().numpy(), np.float32(sc))
        deq = sq.dequantize_payloads({"x": pay}, method)["x"]
        assert bits_equal(deq.cpu().numpy(), deq_ref), (method, bits)
        if method == "asymmetric":      # round trip within half a quantization step
            step = 1.0 / float(sc)
            assert float((deq.cpu() - x).abs().max()) <= 0.5 * step * (1 + 1e-5)


def test_ingest_full_size_and_fused_statistics(sq):
    """ViT-L-14 c_fc.weight x 8 tasks (134 MB of deltas): exact subtraction, base read once; the statistics
    emitted by the ingest pass give the same codes as the separate statistics pass."""
    torch.manual_seed(3)
    D, N = 4096 * 1024, 8
    base = torch.randn(D, device="cuda")
    fts = [base + 0.01 * torch.randn(D, device="cuda") for _ in range(N)]
    batch = sq.ElementwiseBatch([D, 1000], N, "cuda")
    small_base = torch.randn(1000, device="cuda")
    small_ft = [small_base + 0.01 * torch.randn(1000, device="cuda") for _ in range(N)]
    deltas = batch.ingest([base, small_base], fts + small_ft, with_stats=True)
    for t in range(N):
        assert torch.equal(deltas[t], fts[t] - base)
        assert torch.equal(deltas[N + t], small_ft[t] - small_base)
    c1, s1, z1 = batch.quantize(deltas, 4, "asymmetric", stats_ready=True)
    c2, s2, z2 = batch.quantize(deltas, 4, "asymmetric", stats_ready=False)
    assert torch.equal(s1, s2) and torch.equal(z1, z2)
    for a, b in zip(c1, c2):
        assert torch.equal(a, b)
    # per-tensor scale = 15 * (1 / (max - min)) and every code in range
    for i, d in enumerate(deltas):
        want = (1.0 / (d.max() - d.min())) * 15
        assert s1[i].item() == want.item()
        assert int(c1[i].max()) == 15 and int(c1[i].min()) == 0
    # dequantize + base in one pass == dequantize, then add
    plain = batch.dequantize(c1, s1, z1, "asymmetric")
    fused = batch.dequantize(c1, s1, z1, "asymmetric", add=[base, small_base])
    for t in range(N):
        assert torch.equal(fused[t], base + plain[t])
        assert float((plain[t] - deltas[t]).abs().max()) <= 0.5 / s1[t].item() * (1 + 1e-5)
    batch.close()


@pytest.mark.parametrize("N,fp16", [(8, True), (3, False), (20, True), (16, True)])
def test_compress_from_base_equals_ingest_then_compress(sq, N, fp16):
    """svdq_compress_from_base (finetuned - base formed inside the streaming passes) produces exactly the artifacts
    of svdq_ingest followed by svdq_compress."""
    from svdq_amd.pipeline import CompressPlan
    dev = torch.device("cuda", 0)
    sizes = [300000, 777, 70001, 1024 * 96 + 2, 12]
    g = torch.Generator().manual_seed(41)
    base = [torch.randn(D, generator=g).to(dev) for D in sizes]
    deltas = [[d.to(dev) for d in orc.synthetic_deltas(D, N, 590 + i)] for i, D in enumerate(sizes)]
    fts = [[base[p] + deltas[p][t] for t in range(N)] for p in range(len(sizes))]
    batch = sq.ElementwiseBatch(sizes, N, dev)
    ing = batch.ingest(base, [f for fs in fts for f in fs])
    vecs = [ing[p * N:(p + 1) * N] for p in range(len(sizes))]
    kw = dict(energy_threshold=0.9, max_rank=None, center=True, fp16=fp16, low_bits=4, rtvq_stages=2, device=dev,
              unit_rows=1024)
    ref = CompressPlan(sizes, N, **kw)
    ref.run(ref.pointer_table(vecs))
    fb = CompressPlan(sizes, N, **kw)
    btab = torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev)
    fb.run_from_base(fb.pointer_table(fts), btab)
    torch.cuda.synchronize()
    sm, sf = ref.fetch_small(), fb.fetch_small()
    assert torch.equal(fb.small, ref.small)
    for p, D in enumerate(sizes):
        a = ref.basis_tensors(p, int(sm.k[p]), int(sm.r[p]), D)
        b = fb.basis_tensors(p, int(sf.k[p]), int(sf.r[p]), D)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    batch.close()


def test_run_from_checkpoints_matches_task_vector_route(sq, g):
    base, fts = _states(g)
    base = {k: v for k, v in base.items() if v.is_floating_point()}
    fts = {t: {k: v for k, v in sd.items() if k in base and v.shape == base[k].shape} for t, sd in fts.items()}
    fts["C"]["w2"] = fts["A"]["w2"] * 1.01          # every task has every parameter: N >= 3 everywhere
    fts["B"]["b1"] = fts["A"]["b1"] + 0.001
    cfg = sq.SVDHybridConfig(svd_energy_threshold=0.9, svd_max_rank=2)
    tv = sq.compute_task_vectors(base, fts, device="cuda")
    bases1, comp1 = sq.run_basis_and_compress(tv, {}, cfg, "cuda")
    bases2, comp2 = sq.run_basis_and_compress_from_checkpoints(base, fts, cfg, "cuda")
    assert sorted(bases1) == sorted(bases2) and sorted(comp1) == sorted(comp2)
    for name in bases1:
        b1, b2 = bases1[name]["masked"], bases2[name]["masked"]
        assert b1["k"] == b2["k"] and torch.equal(b1["U_high"], b2["U_high"]) and torch.equal(b1["U_low"], b2["U_low"])
        assert torch.equal(b1["mean"], b2["mean"])
        for t in comp1[name]:
            a1, a2 = comp1[name][t]["masked"], comp2[name][t]["masked"]
            assert torch.equal(a1["c_high_fp16"], a2["c_high_fp16"])
            for p1, p2 in zip(a1["c_low_quant"]["payloads"], a2["c_low_quant"]["payloads"]):
                assert torch.equal(p1["quantized"], p2["quantized"])
                assert bits_equal(p1["scale"].numpy(), p2["scale"].numpy())      # NaN == NaN (F4: one-element c_low)
    # masked parameters straight from checkpoints (gather + minus-base in one pass): same as the task-vector route
    gm = torch.Generator().manual_seed(3)
    masks = {"w1": torch.rand(tv[next(iter(tv))]["w1"].shape, generator=gm) > 0.4}
    cfgn = sq.SVDHybridConfig(svd_energy_threshold=0.9, svd_max_rank=2, svd_include_noise=True)
    bases3, comp3 = sq.run_basis_and_compress(tv, masks, cfgn, "cuda")
    bases4, comp4 = sq.run_basis_and_compress_from_checkpoints(base, fts, cfgn, "cuda", combined_masks=masks)
    assert sorted(bases3) == sorted(bases4)
    for name in bases3:
        for region, akey in (("masked", "masked"), ("noise", "unmasked")):
            b3, b4 = bases3[name][region], bases4[name][region]
            assert (b3 is None) == (b4 is None)
            if b3 is None:
                continue
            assert b3["k"] == b4["k"] and b3["D"] == b4["D"]
            assert torch.equal(b3["U_high"], b4["U_high"]) and torch.equal(b3["U_low"], b4["U_low"])
            for t in comp3[name]:
                a3, a4 = comp3[name][t][akey], comp4[name][t][akey]
                assert torch.equal(a3["c_high_fp16"], a4["c_high_fp16"])
                for p3, p4 in zip(a3["c_low_quant"]["payloads"], a4["c_low_quant"]["payloads"]):
                    assert torch.equal(p3["quantized"], p4["quantized"])
    assert bases3["w1"]["noise"] is not None and bases3["w1"]["masked"]["D"] == int(masks["w1"].sum())
