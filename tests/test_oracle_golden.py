"""
Pins the CPU oracle (oracle/) to vectors produced by the reference itself
(tests/golden/*.npz, generator: tests/golden/make_golden.py) and to the reference's own
KATs (reference tests/test_rank_selection.py, test_mask_strategies.py, test_rtvq.py,
test_mean_handling.py re-expressed).  CPU only.
"""
import numpy as np
import pytest
import torch

from oracle import svd_hybrid_oracle as orc
from helpers import load_golden, bits_equal, as_tensors, rel_err

RTVQ_CONFIGS = ((4, 2), (4, 4), (8, 2), (2, 2), (2, 3), (1, 2))


# ------------------------------------------------------------------------------- quantizer
@pytest.fixture(scope="module")
def rtvq_cases():
    return load_golden("rtvq_cases.npz")


def test_rtvq_bit_exact_vs_reference(rtvq_cases):
    """codes / scale / zero_point are bit-identical to the reference for every fixture,
    including the degenerate F4 inputs (scale = inf, NaN) and a NaN-carrying input."""
    g = rtvq_cases
    for name in g["names"]:
        x = g[f"{name}__x"]
        for bits, stages in RTVQ_CONFIGS:
            tag = f"{name}__b{bits}s{stages}__"
            got = orc.rtvq_quantize(x, bits, stages)
            assert np.array_equal(got["codes"], g[tag + "codes"]), tag
            assert bits_equal(got["scale"], g[tag + "scale"]), tag
            assert bits_equal(got["zero_point"], g[tag + "zero_point"]), tag
            np.testing.assert_allclose(got["residual_norm"], g[tag + "residual_norm"], rtol=1e-5,
                                       equal_nan=True, err_msg=tag)
            assert bits_equal(orc.rtvq_dequantize(got), g[tag + "deq"]), tag
            # second (numpy) statement agrees with the C build
            alt = orc.rtvq_quantize_numpy(x, bits, stages)
            assert np.array_equal(alt["codes"], got["codes"]), tag
            assert bits_equal(alt["scale"], got["scale"]) and bits_equal(alt["zero_point"], got["zero_point"])


def test_single_stage_functions(rtvq_cases):
    g = rtvq_cases
    for name in g["names"]:
        x = g[f"{name}__x"]
        for bits in (8, 4, 2):
            tag = f"{name}__asym{bits}__"
            q, sc, zp = orc.asym_quantize(x, bits)
            assert q.dtype == np.uint8 and np.array_equal(q, g[tag + "q"])
            assert bits_equal(np.float32(sc), g[tag + "scale"])
            assert bits_equal(np.float32(zp), g[tag + "zero_point"])
            assert bits_equal(orc.asym_dequantize(q, sc, zp), g[tag + "deq"])


def test_rtvq_empty_and_f4(rtvq_cases):
    empty = orc.rtvq_quantize(np.zeros(0, np.float32), 4, 2)
    assert empty["codes"].size == 0 and orc.rtvq_dequantize(empty).size == 0
    assert int(rtvq_cases["empty__deq_numel"]) == 0
    one = orc.rtvq_quantize(np.array([0.5], np.float32), 4, 2)
    assert np.isinf(one["scale"][0]) and one["codes"][0, 0] == 0
    assert np.isnan(orc.rtvq_dequantize(one)).all()


def test_rtvq_reference_kats():
    """reference tests/test_rtvq.py:33-46 (error bound), :68-84, :112-133 (monotone)."""
    x = np.array([1, 2, 3, 4, 5], np.float32)
    q, sc, zp = orc.asym_quantize(x, 4)
    assert np.abs(x - orc.asym_dequantize(q, sc, zp)).max() <= sc / 2 * 1.5
    rng = np.random.default_rng(0)
    y = rng.standard_normal(100).astype(np.float32)
    errs = [np.linalg.norm(y - orc.rtvq_dequantize(orc.rtvq_quantize(y, 2, s))) for s in (1, 2, 3)]
    assert errs[0] > errs[1] > errs[2]
    eb = [np.linalg.norm(y - orc.rtvq_dequantize(orc.rtvq_quantize(y, b, 1))) for b in (2, 4, 8)]
    assert eb[0] > eb[1] > eb[2]


def test_rtvq_large_config1():
    """BASELINE.json configs[0] quantizer leg: 768x768, seeded; histograms + checksums."""
    g = load_golden("rtvq_large.npz")
    torch.manual_seed(int(g["seed"]))
    x = (0.01 * torch.randn(768, 768)).numpy()
    assert bits_equal(x.ravel()[:8], g["x_first8"])
    w = (np.arange(x.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    for bits, stages in ((4, 2), (8, 2), (2, 4)):
        tag = f"b{bits}s{stages}__"
        got = orc.rtvq_quantize(x, bits, stages)
        assert bits_equal(got["scale"], g[tag + "scale"])
        assert bits_equal(got["zero_point"], g[tag + "zero_point"])
        np.testing.assert_allclose(got["residual_norm"], g[tag + "residual_norm"], rtol=1e-5)
        for s in range(stages):
            assert np.array_equal(np.bincount(got["codes"][s], minlength=256), g[tag + "hist"][s])
            assert int((got["codes"][s].astype(np.uint64) * w).sum()) == int(g[tag + "weighted_sum"][s])
        assert bits_equal(orc.rtvq_dequantize(got).ravel()[:64], g[tag + "deq_head"])


# ------------------------------------------------------------------------------- rank selection
def test_rank_kats_vs_reference():
    g = load_golden("rank_kats.npz")
    names = sorted({k.split("__")[0] for k in g})
    for name in names:
        S = torch.from_numpy(g[f"{name}__S"])
        assert bits_equal(orc.energy_spectrum(S).numpy(), g[f"{name}__cum"])
        for thr in (0.5, 0.9, 0.95, 0.99, 0.999, 1.0):
            for mr in (None, 2, 10):
                assert orc.select_rank(S, thr, mr) == int(g[f"{name}__k__thr{thr}__mr{mr}"])
        assert orc.select_rank(S, 0.999, None, min_rank=2) == int(g[f"{name}__k__minrank2"])
    # reference tests/test_rank_selection.py:58-65, :27-33, :36-42
    assert orc.select_rank(torch.tensor([10.0, 1e-10, 1e-12]), 0.99) == 1
    assert orc.select_rank(torch.ones(100), 0.99, max_rank=10) <= 10
    assert orc.select_rank(torch.tensor([100.0, 0.01, 0.001]), 0.999, None, min_rank=2) >= 2


# ------------------------------------------------------------------------------- basis chain
BASIS_FIXTURES = ["basis_d768_n8", "basis_d768_n3", "basis_d768_n20", "basis_d768_n20b",
                  "basis_d4096_n8", "basis_d4096_n8_nocenter", "basis_d4096_n8_fp32",
                  "basis_d1000_n5", "basis_d999_n12"]


def _deltas_for(g):
    if "deltas" in g:
        return as_tensors(g["deltas"])
    ds = orc.synthetic_deltas(int(g["D"]), int(g["N"]), int(g["seed"]))
    assert abs(torch.stack(ds).double().sum().item() - float(g["delta_sum"])) < 1e-9
    return ds


SPECTRUM_FIXTURES = ["spectrum_graded_n8", "spectrum_graded_n8c", "spectrum_graded_n16", "spectrum_graded_n20", "spectrum_rankdef_n6",
                     "spectrum_twins_n6c", "spectrum_thresh_below_n8", "spectrum_thresh_above_n8", "spectrum_gap_n20a",
                     "spectrum_gap_n20b"]


@pytest.mark.parametrize("name", BASIS_FIXTURES + ["basis_d65536_n8"] + SPECTRUM_FIXTURES)
def test_basis_chain_vs_reference(name):
    """Same machine, same LAPACK: the oracle must reproduce the reference's chain exactly
    for integers and to BLAS-reduction-order tolerance for floats."""
    g = load_golden(name + ".npz")
    deltas = _deltas_for(g)
    mr = None if int(g["max_rank"]) < 0 else int(g["max_rank"])
    res = orc.compress_parameter(deltas, float(g["thr"]), mr, bool(g["center"]), bool(g["fp16"]),
                                 int(g["bits"]), int(g["stages"]))
    b = res["basis"]
    assert b["k"] == int(g["k"]) and b["N"] == int(g["N"]) and b["D"] == int(g["D"])
    np.testing.assert_allclose(b["singular_values"].numpy(), g["S"], rtol=1e-6, atol=1e-9)
    assert abs(b["energy_retained"] - float(g["energy_retained"])) < 1e-6
    if bool(g["center"]):
        np.testing.assert_allclose(b["mean"].squeeze(1).numpy()[:64], g["mean_head"], rtol=1e-6, atol=1e-10)
    else:
        assert b["mean"] is None
    if "U_high" in g:
        assert b["U_high"].shape == g["U_high"].shape and b["U_low"].shape == g["U_low"].shape
        np.testing.assert_allclose(b["U_high"].float().numpy(), g["U_high"].astype(np.float32), atol=2e-3)
    for t, art in enumerate(res["tasks"]):
        assert np.array_equal(art["c_high_fp16"].numpy().view(np.uint16),
                              g["c_high_fp16"][t].view(np.uint16))
        q = art["c_low_quant"]
        assert np.array_equal(q["codes"], g[f"t{t}__codes"])
        assert bits_equal(q["scale"], g[f"t{t}__scale"])
        assert bits_equal(q["zero_point"], g[f"t{t}__zero_point"])
        np.testing.assert_allclose(q["residual_norm"], g[f"t{t}__residual_norm"], rtol=1e-5)
    recon = np.stack([r.numpy() for r in res["recon"]])
    if "recon" in g:
        assert np.mean((recon - g["recon"]) ** 2) <= 1e-12
    np.testing.assert_allclose(recon[:, :64], g["recon_head"], rtol=1e-4, atol=1e-7)


def test_config1_plumbing():
    """configs[0]: N=2, 768x768; center=False -> k=2, empty U_low, no payloads;
    center=True -> the reference's F4 NaN, reproduced not hidden."""
    g = load_golden("config1.npz")
    torch.manual_seed(0)
    deltas = [0.01 * torch.randn(768 * 768) for _ in range(2)]
    assert abs(torch.stack(deltas).double().sum().item() - float(g["delta_sum"])) < 1e-9
    nc = orc.compress_parameter(deltas, 0.9, None, False, True, 4, 2)
    assert nc["basis"]["k"] == int(g["nocenter__k"]) == 2
    assert nc["basis"]["U_low"].shape == (768 * 768, 0)
    assert all(t["c_low_quant"]["codes"].size == 0 for t in nc["tasks"])
    assert int(g["nocenter__t0__n_payloads"]) == 0
    np.testing.assert_allclose(nc["basis"]["singular_values"].numpy(), g["nocenter__S"], rtol=1e-6)
    rel = [rel_err(r.numpy(), d.numpy()) for r, d in zip(nc["recon"], deltas)]
    np.testing.assert_allclose(rel, g["nocenter__recon_rel_err"], rtol=1e-2)
    ce = orc.compress_parameter(deltas, 0.9, None, True, True, 4, 2)
    assert ce["basis"]["k"] == int(g["center__k"]) == 1
    assert np.isinf(ce["tasks"][0]["c_low_quant"]["scale"][0]) and np.isinf(g["center__t0__scale"][0])
    assert all(np.isnan(r.numpy()).all() for r in ce["recon"])
    assert np.isnan(g["center__recon_rel_err"]).all()


# ------------------------------------------------------------------------------- masks
def test_masks_vs_reference():
    g = load_golden("masks.npz")
    masks = [torch.from_numpy(m) for m in g["masks"]]
    for strat in ("union", "intersection", "majority"):
        assert np.array_equal(orc.combine_masks(masks, strat).numpy(), g[f"combined_{strat}"])
    assert np.array_equal(orc.combine_masks(masks[:4], "majority").numpy(), g["majority_even4"])
    union = torch.from_numpy(g["combined_union"])
    deltas = [torch.from_numpy(d) for d in g["deltas"]]
    sig = [orc.select_masked(d, union) for d in deltas]
    noi = [orc.select_unmasked(d, union) for d in deltas]
    assert np.array_equal(torch.stack(sig).numpy(), g["signal"])
    assert np.array_equal(torch.stack(noi).numpy(), g["noise"])
    assert torch.equal(orc.scatter_masked(sig[0], noi[0], union, deltas[0].shape), deltas[0])
    assert np.array_equal(orc.scatter_masked(sig[1], None, union, deltas[1].shape).numpy(),
                          g["scatter_signal_only"])
    for region, vecs in (("masked", sig), ("noise", noi)):
        b = orc.svd_basis(vecs, 0.9, None, True)
        assert b["k"] == int(g[f"{region}__k"]) and b["D"] == int(g[f"{region}__D"])
        np.testing.assert_allclose(b["singular_values"].numpy(), g[f"{region}__S"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(b["mean"].squeeze(1).numpy(), g[f"{region}__mean"], rtol=1e-6, atol=1e-10)


def test_mask_reference_kats():
    """reference tests/test_mask_strategies.py:18-135 (exact outputs incl. even-N tie),
    :138-257 (empty list, single task, shape mismatch)."""
    a = torch.tensor([[True, False, True], [False, False, True]])
    b = torch.tensor([[False, False, True], [True, False, True]])
    c = torch.tensor([[False, True, True], [False, False, False]])
    assert torch.equal(orc.combine_masks([a, b, c], "union"),
                       torch.tensor([[True, True, True], [True, False, True]]))
    assert torch.equal(orc.combine_masks([a, b, c], "intersection"),
                       torch.tensor([[False, False, True], [False, False, False]]))
    assert torch.equal(orc.combine_masks([a, b, c], "majority"),
                       torch.tensor([[False, False, True], [False, False, True]]))
    assert torch.equal(orc.combine_masks([a, b], "majority"), a | b)       # tie counts: 1 >= 0.5*2
    assert torch.equal(orc.combine_masks([a], "union"), a)
    for strat in ("union", "intersection", "majority"):
        with pytest.raises(ValueError):
            orc.combine_masks([], strat)
    with pytest.raises(ValueError):
        orc.combine_masks([a], "nope")
    with pytest.raises(ValueError):
        orc.select_masked(torch.zeros(2, 3), torch.zeros(3, 2, dtype=torch.bool))
    t = torch.tensor([[1, 2, 3], [4, 5, 6]])
    m = torch.tensor([[True, False, True], [False, True, False]])
    assert orc.select_masked(t, m).tolist() == [1, 3, 5]
    assert orc.select_unmasked(t, m).tolist() == [2, 4, 6]


# ------------------------------------------------------------------------------- reconstruction
def test_parameter_diagnostics_vs_reference_artifacts():
    """The oracle's restatement of diagnostics.py:72-117,186-215 on the reference's OWN artifacts (diag.npz holds the
    basis, coefficients and masked originals the reference computed its numbers from): in fp32 it is the reference's
    arithmetic (equal to the last bit of the printed doubles); in fp64 it stays inside the stated fp32 forward-error
    bound -- the bound the GPU tests then hold the HIP kernels to, on the kernels' own artifacts."""
    from helpers import diag_fp32_bound
    g = load_golden("diag.npz")
    worst = 0.0
    for c in g["cases"]:
        for p in g[f"{c}__params"]:
            X = torch.from_numpy(g[f"{c}__x__{p}"])
            Uh, Ul = torch.from_numpy(g[f"{c}__U_high__{p}"]), torch.from_numpy(g[f"{c}__U_low__{p}"])
            ch, cl = torch.from_numpy(g[f"{c}__c_high_fp16__{p}"]).float(), torch.from_numpy(g[f"{c}__c_low_deq__{p}"])
            M = g[f"{c}__metrics__{p}"]
            assert Uh.shape[0] == X.shape[1] and Uh.shape[1] + Ul.shape[1] == X.shape[0]
            for t in range(X.shape[0]):
                m32 = orc.parameter_task_diagnostics(X[t], Uh, Ul, ch[t], cl[t])
                m64 = orc.parameter_task_diagnostics(X[t], Uh, Ul, ch[t], cl[t], dtype=torch.float64)
                tol = diag_fp32_bound(Uh, Ul, ch[t], cl[t], X[t])
                for j, key in enumerate(orc.DIAG_KEYS):
                    assert m32[key] == M[t, j], (c, p, t, key)
                    assert abs(m64[key] - M[t, j]) <= 2e-6 * abs(M[t, j]) + tol[key], (c, p, t, key)
                    worst = max(worst, abs(m64[key] - M[t, j]) / abs(M[t, j]))
    assert 1e-5 < worst < 2e-4      # the reference's fp32 numbers ARE further than 1e-5 from fp64 (max_absolute_error)


def test_reconstruction_identity_kat():
    """reference tests/test_mean_handling.py:206-312: reconstruct == U_h c_h + U_l c_l (+mean)
    and projection round trip, to 1e-5, for a random orthonormal U."""
    torch.manual_seed(42)
    D, N, k = 100, 4, 2
    Q, _ = torch.linalg.qr(torch.randn(D, N))
    U_high, U_low = Q[:, :k].contiguous(), Q[:, k:].contiguous()
    c_high, c_low, mean = torch.randn(k), torch.randn(N - k), torch.randn(D, 1)
    want = U_high @ c_high + U_low @ c_low
    assert torch.allclose(orc.reconstruct(c_high, c_low, U_high, U_low, None), want, atol=1e-5)
    assert torch.allclose(orc.reconstruct(c_high, c_low, U_high, U_low, mean), want + mean.squeeze(), atol=1e-5)
    ch, cl = orc.project(want, U_high, U_low)
    assert torch.allclose(ch, c_high, atol=1e-5) and torch.allclose(cl, c_low, atol=1e-5)


def test_empty_inputs_raise():
    with pytest.raises(ValueError):
        orc.stack_center([], True)
    with pytest.raises(ValueError):
        orc.svd_basis([])
