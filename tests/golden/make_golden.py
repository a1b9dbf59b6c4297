#!/usr/bin/env python3
"""
tests/golden/make_golden.py -- regenerates tests/golden/*.npz by running the REFERENCE itself.

Run only in the build container, where /root/reference is mounted (it does not exist on
the GPU box; nothing under tests/ reads it at test time):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference package ``src.svd_hybrid`` cannot be imported normally here (its __init__
pulls torchvision/open_clip), so empty package modules are pre-registered and
only the hot-path submodules are imported.  The fixtures are DATA:
seeded inputs and the reference's outputs for them.  No reference source is stored.

Fixture families (all .npz, loadable with allow_pickle=False):
  rtvq_*.npz      quantizer: rtvq.py asymmetric/multistage functions and RTVQQuantizer
  rtvq_large.npz  config #1: RTVQQuantizer(4,2) on a 768x768 tensor (seeded; checksums)
  basis_*.npz     construct_basis -> .half() -> compress_single_task -> dequantize ->
                  reconstruct_from_coefficients (the four-call parity chain)
  spectrum_*.npz  the same chain on inputs whose spectrum reaches down to 3e-6 sigma_0, on exactly dependent tasks
                  and with the cumulative energy 5e-5 below / above the threshold (with fp64 singular values beside)
  config1.npz     N=2, one [768,768], center False/True (the F4 NaN case recorded as such)
  masks.npz       combine_masks / apply_mask_to_tensor / get_unmasked_portion /
                  reconstruct_from_masked / construct_masked_basis(include_noise)
  pipeline.npz    cli.py Step 4 + Step 5 loop bodies over 3 layers x 4 tasks
                  (construct_masked_basis + compress_all_parameters), dict layout + numbers
  merge.npz       merge_all_parameters + apply_merged_deltas (merge.py:304-552), 6 tasks, weighted
  diag.npz        compute_all_diagnostics' per-(parameter, task) error tuples TOGETHER WITH the artifacts they were
                  computed from (the reference's own U_high / U_low / c_high_fp16 / dequantized c_low and the masked
                  originals), so a checker can be pinned on them with no basis freedom left (diagnostics.py:186-215)
  tvq.npz         TaskVector / compute_task_vector and the QuantizedTaskVector family
                  (task_vectors.py, quantization_utils.py) on a toy state dict x 3 tasks
  cluster.npz     cluster_tasks / compute_cluster_statistics (clustering.py), compute_weights family
                  (weighting.py), merge_with_clustering (merge.py:555-626) on merge.npz's inputs
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
for _name, _path in (("src", REF + "/src"), ("src.svd_hybrid", REF + "/src/svd_hybrid")):
    _m = types.ModuleType(_name)
    _m.__path__ = [_path]
    sys.modules[_name] = _m

from src.svd_hybrid import rtvq as ref_rtvq                      # noqa: E402
from src.svd_hybrid import basis as ref_basis                    # noqa: E402
from src.svd_hybrid import compress as ref_compress              # noqa: E402
from src.svd_hybrid import merge as ref_merge                    # noqa: E402
from src.svd_hybrid import mask_loader as ref_masks              # noqa: E402
from src.svd_hybrid import diagnostics as ref_diag               # noqa: E402
from src.svd_hybrid import clustering as ref_cluster             # noqa: E402
from src.svd_hybrid import weighting as ref_weighting            # noqa: E402
import quantization_utils as ref_qutils                          # noqa: E402
import task_vectors as ref_tv                                    # noqa: E402
from src.svd_hybrid import task_vector_loader as ref_loader      # noqa: E402

sys.path.insert(0, ROOT)
from oracle.svd_hybrid_oracle import synthetic_deltas            # noqa: E402  (input generator only)

torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def payload_arrays(prefix, qobj, out):
    """Flatten RTVQQuantizer.quantize output into arrays under ``prefix``."""
    p = qobj["payloads"]
    out[prefix + "n_payloads"] = np.int64(len(p))
    if p:
        out[prefix + "codes"] = np.stack([x["quantized"].numpy() for x in p])
        out[prefix + "scale"] = np.array([x["scale"].item() for x in p], dtype=np.float32)
        out[prefix + "zero_point"] = np.array([x["zero_point"].item() for x in p], dtype=np.float32)
        out[prefix + "residual_norm"] = np.array([x["residual_norm"] for x in p], dtype=np.float64)
        assert [x["stage"] for x in p] == list(range(len(p)))
        assert all(x["quantized"].dtype == torch.uint8 for x in p)
        assert all(x["scale"].ndim == 0 and x["zero_point"].ndim == 0 for x in p)


# ------------------------------------------------------------------------------- quantizer
def gen_rtvq():
    cases = {}
    g = torch.Generator().manual_seed(1234)
    inputs = {
        "n7": torch.randn(7, generator=g),
        "n19": torch.randn(19, generator=g) * 3.0 + 1.0,
        "n2": torch.tensor([-0.25, 0.75]),
        "n100": torch.randn(100, generator=g),
        "n4096": torch.randn(4096, generator=g) * 0.01,
        "ramp5": torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0]),          # reference tests/test_rtvq.py:35
        "ties": torch.tensor([0.0, 0.5, 1.5, 2.5, 3.5, 7.5, 15.0]),  # half-way cases at 4 bits
        "one": torch.tensor([0.5]),                                 # F4: scale = inf
        "const": torch.ones(5) * 3.14,                              # F4
        "zeros": torch.zeros(3),                                    # F4
        "hasnan": torch.tensor([1.0, float("nan"), -2.0, 0.5]),
    }
    for name, x in inputs.items():
        cases[f"{name}__x"] = x.numpy()
        for bits, stages in ((4, 2), (4, 4), (8, 2), (2, 2), (2, 3), (1, 2)):
            tag = f"{name}__b{bits}s{stages}__"
            quant = ref_rtvq.RTVQQuantizer(num_bits=bits, num_stages=stages)
            obj = quant.quantize(x)
            payload_arrays(tag, obj, cases)
            cases[tag + "deq"] = quant.dequantize(obj).numpy()
            assert obj["num_bits"] == bits and obj["num_stages"] == stages
            assert tuple(obj["original_shape"]) == tuple(x.shape)
            assert obj["original_dtype"] == "torch.float32"
        # single-stage function + the quantization_utils copy named by north_star
        for bits in (8, 4, 2):
            q, sc, zp = ref_rtvq.asymmetric_quantization(x, bits)
            q2, sc2, zp2 = ref_qutils.asymmetric_quantization(x, bits)
            assert torch.equal(q, q2)
            tag = f"{name}__asym{bits}__"
            cases[tag + "q"] = q.numpy()
            cases[tag + "scale"] = np.float32(sc.item())
            cases[tag + "zero_point"] = np.float32(zp.item())
            cases[tag + "deq"] = ref_rtvq.asymmetric_dequantization(q, sc, zp).numpy()
            d2 = ref_qutils.dequantize_asymmetric(q2, sc2, zp2).numpy()
            assert np.array_equal(cases[tag + "deq"], d2, equal_nan=True)
    # qbit = 16 (rtvq.py:22-25): int16 codes straight from the reference's cast (values above 32767 wrap)
    for name in ("n19", "n100", "n4096", "ramp5"):
        q, sc, zp = ref_rtvq.asymmetric_quantization(inputs[name], 16)
        assert q.dtype == torch.int16
        tag = f"{name}__asym16__"
        cases[tag + "q"] = q.numpy()
        cases[tag + "scale"] = np.float32(sc.item())
        cases[tag + "zero_point"] = np.float32(zp.item())
        cases[tag + "deq"] = ref_rtvq.asymmetric_dequantization(q, sc, zp).numpy()
    empty = ref_rtvq.RTVQQuantizer(4, 2).quantize(torch.tensor([]))
    assert empty["payloads"] == []
    cases["empty__deq_numel"] = np.int64(ref_rtvq.RTVQQuantizer(4, 2).dequantize(empty).numel())
    cases["names"] = np.array(list(inputs.keys()))
    save("rtvq_cases.npz", **cases)


def gen_rtvq_large():
    """Config #1: RTVQQuantizer(4,2) directly on the 589,824-element tensor."""
    torch.manual_seed(0)
    x = 0.01 * torch.randn(768, 768)
    out = {"seed": np.int64(0), "x_sum": np.float64(x.double().sum().item()),
           "x_first8": x.flatten()[:8].numpy()}
    for bits, stages in ((4, 2), (8, 2), (2, 4)):
        quant = ref_rtvq.RTVQQuantizer(bits, stages)
        obj = quant.quantize(x)
        tag = f"b{bits}s{stages}__"
        p = obj["payloads"]
        out[tag + "scale"] = np.array([t["scale"].item() for t in p], dtype=np.float32)
        out[tag + "zero_point"] = np.array([t["zero_point"].item() for t in p], dtype=np.float32)
        out[tag + "residual_norm"] = np.array([t["residual_norm"] for t in p], dtype=np.float64)
        out[tag + "hist"] = np.stack([np.bincount(t["quantized"].numpy().ravel(), minlength=256)
                                      for t in p])
        w = (np.arange(x.numel(), dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
        out[tag + "weighted_sum"] = np.array(
            [int((t["quantized"].numpy().ravel().astype(np.uint64) * w).sum()) for t in p],
            dtype=np.uint64)
        out[tag + "codes_head"] = np.stack([t["quantized"].numpy().ravel()[:64] for t in p])
        deq = quant.dequantize(obj)
        assert deq.shape == x.shape
        out[tag + "deq_head"] = deq.flatten()[:64].numpy()
        out[tag + "rel_err"] = np.float64(((x - deq).norm() / x.norm()).item())
    save("rtvq_large.npz", **out)


# ------------------------------------------------------------------------------- basis chain
def basis_chain(deltas, thr, max_rank, center, fp16, bits, stages, store_inputs=True, store_full=True):
    N = len(deltas)
    b = ref_basis.construct_basis(deltas, energy_threshold=thr, max_rank=max_rank, center=center,
                                  device="cpu", verbose=False)
    out = {
        "N": np.int64(N), "D": np.int64(b["D"]), "thr": np.float64(thr),
        "max_rank": np.int64(-1 if max_rank is None else max_rank), "center": np.bool_(center),
        "fp16": np.bool_(fp16), "bits": np.int64(bits), "stages": np.int64(stages),
        "S": b["singular_values"].numpy(), "k": np.int64(b["k"]),
        "energy_retained": np.float64(b["energy_retained"]),
    }
    assert b["N"] == N
    if store_inputs:
        out["deltas"] = torch.stack(deltas).numpy()
    U_high, U_low = b["U_high"], b["U_low"]
    assert U_high.is_contiguous() and U_low.is_contiguous()
    assert U_high.shape == (b["D"], b["k"]) and U_low.shape == (b["D"], N - b["k"])
    if center:
        assert b["mean"].shape == (b["D"], 1)
        m = b["mean"].squeeze(1)
        if store_full or b["D"] <= 65536:
            out["mean"] = m.numpy()
        out["mean_sum"] = np.float64(m.double().sum().item())
        out["mean_head"] = m[:64].numpy()
    else:
        assert b["mean"] is None
    if fp16:                                                   # cli.py:354-361
        U_high, U_low = U_high.half(), U_low.half()
    if store_full:
        out["U_high"] = U_high.numpy()
        out["U_low"] = U_low.numpy()
    quant = ref_rtvq.RTVQQuantizer(bits, stages)
    c_high_all, c_low_all, c16_all, deq_all, recon_all = [], [], [], [], []
    for t, d in enumerate(deltas):
        d_c = d if b["mean"] is None else d - b["mean"].squeeze()
        ch, cl = ref_compress.project_to_basis(d_c, U_high, U_low)
        art = ref_compress.compress_single_task(d, U_high, U_low, quant, "cpu", mean=b["mean"])
        assert set(art.keys()) == {"c_high_fp16", "c_low_quant"}
        assert art["c_high_fp16"].dtype == torch.float16
        payload_arrays(f"t{t}__", art["c_low_quant"], out)
        cl_hat = quant.dequantize(art["c_low_quant"])
        rec = ref_merge.reconstruct_from_coefficients(art["c_high_fp16"].float(), cl_hat.float(),
                                                      U_high, U_low, "cpu", mean=b["mean"])
        c_high_all.append(ch.numpy()); c_low_all.append(cl.numpy())
        c16_all.append(art["c_high_fp16"].numpy()); deq_all.append(cl_hat.numpy().reshape(-1))
        recon_all.append(rec.numpy())
    out["c_high"] = np.stack(c_high_all)
    out["c_low"] = np.stack(c_low_all)
    out["c_high_fp16"] = np.stack(c16_all)
    out["c_low_deq"] = np.stack(deq_all)
    recon = np.stack(recon_all)
    if store_full:
        out["recon"] = recon
    orig = torch.stack(deltas).numpy()
    out["recon_rel_err"] = np.linalg.norm(recon - orig, axis=1) / np.linalg.norm(orig, axis=1)
    out["recon_sq_sum"] = (recon.astype(np.float64) ** 2).sum(axis=1)
    out["recon_head"] = recon[:, :64].copy()
    return out


def gen_basis():
    specs = [
        # name,            D,     N,  seed, thr,  max_rank, center, fp16, bits, stages, full
        ("basis_d768_n8",   768,   8,  11,   0.90, None,     True,   True, 4,    2,      True),
        ("basis_d768_n3",   768,   3,  12,   0.90, None,     True,   True, 4,    2,      True),
        ("basis_d768_n20",  768,   20, 13,   0.90, None,     True,   True, 8,    2,      True),
        ("basis_d768_n20b", 768,   20, 13,   0.95, 64,       True,   True, 2,    2,      True),
        ("basis_d4096_n8",  4096,  8,  14,   0.90, None,     True,   True, 4,    4,      True),
        ("basis_d4096_n8_nocenter", 4096, 8, 14, 0.90, None, False,  True, 4,    2,      True),
        ("basis_d4096_n8_fp32",     4096, 8, 14, 0.95, 64,   True,   False, 4,   2,      True),
        ("basis_d1000_n5",  1000,  5,  15,   0.50, None,     True,   True, 4,    2,      True),
        ("basis_d999_n12",  999,   12, 16,   0.99, 4,        True,   True, 4,    2,      True),
        ("basis_d65536_n8", 65536, 8,  17,   0.90, None,     True,   True, 4,    2,      False),
    ]
    for name, D, N, seed, thr, mr, center, fp16, bits, stages, full in specs:
        deltas = synthetic_deltas(D, N, seed)
        out = basis_chain(deltas, thr, mr, center, fp16, bits, stages,
                          store_inputs=full, store_full=full)
        out["seed"] = np.int64(seed)
        out["delta_sum"] = np.float64(torch.stack(deltas).double().sum().item())
        save(name + ".npz", **out)


def gen_config1():
    """BASELINE.json configs[0]: 2 synthetic tasks, single 768x768 linear, 2-stage 4-bit."""
    torch.manual_seed(0)
    deltas = [0.01 * torch.randn(768 * 768) for _ in range(2)]
    out = {"delta_sum": np.float64(torch.stack(deltas).double().sum().item())}
    for center in (False, True):
        tag = "center__" if center else "nocenter__"
        r = basis_chain(deltas, 0.9, None, center, True, 4, 2, store_inputs=False, store_full=False)
        for key, val in r.items():
            out[tag + key] = val
    save("config1.npz", **out)


# ------------------------------------------------------------------------------- masks
def gen_masks():
    out = {}
    g = torch.Generator().manual_seed(77)
    shape = (40, 50)
    N = 5
    masks = [torch.rand(shape, generator=g) > 0.7 for _ in range(N)]
    deltas = [d.view(shape) for d in synthetic_deltas(shape[0] * shape[1], N, 21)]
    out["masks"] = torch.stack(masks).numpy()
    out["deltas"] = torch.stack(deltas).numpy()
    task_masks = {f"t{i}": {"w": m} for i, m in enumerate(masks)}
    task_masks["none_task"] = None                                   # mask_loader.py:589-594
    for strat in ("union", "intersection", "majority"):
        comb = ref_masks.combine_masks(task_masks, strategy=strat, device="cpu", verbose=False)
        out[f"combined_{strat}"] = comb["w"].numpy()
    # even-N tie for majority (reference tests/test_mask_strategies.py): >= 0.5 * n
    even = ref_masks.compute_majority_mask(masks[:4])
    out["majority_even4"] = even.numpy()
    # other thresholds (mask_loader.py:456-485), incl. products that are not exact in fp32 and the two extremes
    thr_list = [0.0, 0.2, 0.3, 0.4, 0.6, 0.75, 0.8, 1.0, 1.2]
    out["majority_thresholds"] = np.array(thr_list, dtype=np.float64)
    for i, thr in enumerate(thr_list):
        out[f"majority_thr{i}_n5"] = ref_masks.compute_majority_mask(masks, threshold=thr).numpy()
        out[f"majority_thr{i}_n3"] = ref_masks.compute_majority_mask(masks[:3], threshold=thr).numpy()
    union = torch.from_numpy(out["combined_union"])
    sig = [ref_masks.apply_mask_to_tensor(d, union) for d in deltas]
    noi = [ref_masks.get_unmasked_portion(d, union) for d in deltas]
    out["signal"] = torch.stack(sig).numpy()
    out["noise"] = torch.stack(noi).numpy()
    back = ref_masks.reconstruct_from_masked(sig[0], noi[0], union, deltas[0].shape)
    assert torch.equal(back, deltas[0])
    out["scatter_signal_only"] = ref_masks.reconstruct_from_masked(sig[1], None, union,
                                                                   deltas[1].shape).numpy()
    mb = ref_basis.construct_masked_basis(sig, noi, energy_threshold=0.9, max_rank=None, center=True,
                                          device="cpu", include_noise=True, verbose=False)
    for region in ("masked", "noise"):
        b = mb[region]
        out[f"{region}__S"] = b["singular_values"].numpy()
        out[f"{region}__k"] = np.int64(b["k"])
        out[f"{region}__energy"] = np.float64(b["energy_retained"])
        out[f"{region}__D"] = np.int64(b["D"])
        out[f"{region}__mean"] = b["mean"].squeeze(1).numpy()
    none_case = ref_basis.construct_masked_basis([], None, verbose=False)
    assert none_case == {"masked": None, "noise": None}
    save("masks.npz", **out)


# ------------------------------------------------------------------------------- pipeline
def gen_pipeline():
    """cli.py:317-361 + compress.py:173-207 over a toy model; pins dict layout and numbers."""
    shapes = {"blk.0.attn.weight": (48, 32), "blk.0.attn.bias": (48,), "blk.1.mlp.weight": (64, 48)}
    tasks = ["Cars", "DTD", "EuroSAT", "GTSRB"]
    cfg = types.SimpleNamespace(svd_low_bits=4, svd_rtvq_stages=2, svd_include_noise=True,
                                svd_min_mask_size=10, svd_energy_threshold=0.9, svd_max_rank=64,
                                svd_center=True, svd_fp16=True)
    g = torch.Generator().manual_seed(5)
    task_vectors = {t: {} for t in tasks}
    masks = {}
    out = {}
    for pi, (pname, shp) in enumerate(sorted(shapes.items())):
        numel = int(np.prod(shp))
        ds = synthetic_deltas(numel, len(tasks), 100 + pi)
        for t, d in zip(tasks, ds):
            task_vectors[t][pname] = d.view(shp)
        out[f"in__{pname}"] = torch.stack(ds).numpy()
        if pname.endswith("weight"):
            per_task = [torch.rand(shp, generator=g) > 0.6 for _ in tasks]
            masks[pname] = ref_masks.compute_union_mask(per_task)
            out[f"mask__{pname}"] = masks[pname].numpy()
    # Step 4 body (cli.py:317-361)
    bases = {}
    for pname in sorted(shapes):
        mask = masks.get(pname)
        md, ud = [], []
        for t in tasks:
            delta = task_vectors[t][pname]
            if mask is not None and mask.shape == delta.shape:
                if mask.sum() >= cfg.svd_min_mask_size:
                    md.append(ref_masks.apply_mask_to_tensor(delta, mask))
                    if cfg.svd_include_noise:
                        ud.append(ref_masks.get_unmasked_portion(delta, mask))
            else:
                md.append(delta.flatten())
        basis = ref_basis.construct_masked_basis(md, ud if cfg.svd_include_noise else None,
                                                 energy_threshold=cfg.svd_energy_threshold,
                                                 max_rank=cfg.svd_max_rank, center=cfg.svd_center,
                                                 device="cpu", include_noise=cfg.svd_include_noise)
        for region in ("masked", "noise"):
            if basis.get(region) is not None:
                basis[region]["U_high"] = basis[region]["U_high"].half()
                basis[region]["U_low"] = basis[region]["U_low"].half()
        bases[pname] = basis
    compressed = ref_compress.compress_all_parameters(task_vectors, masks, bases, cfg, device="cpu")
    layout = {}
    quant = ref_rtvq.RTVQQuantizer(cfg.svd_low_bits, cfg.svd_rtvq_stages)
    for pname in sorted(compressed):
        layout[pname] = {}
        for region in ("masked", "noise"):
            b = bases[pname][region]
            layout[pname][f"basis_{region}"] = None if b is None else sorted(b.keys())
            if b is not None:
                out[f"basis__{pname}__{region}__S"] = b["singular_values"].numpy()
                out[f"basis__{pname}__{region}__k"] = np.int64(b["k"])
                out[f"basis__{pname}__{region}__D"] = np.int64(b["D"])
                out[f"basis__{pname}__{region}__energy"] = np.float64(b["energy_retained"])
        for t in tasks:
            art = compressed[pname][t]
            layout[pname][t] = {r: (None if art[r] is None else sorted(art[r].keys()))
                                for r in sorted(art.keys())}
            for region, bkey in (("masked", "masked"), ("unmasked", "noise")):
                a = art[region]
                if a is None:
                    continue
                tag = f"coef__{pname}__{t}__{region}__"
                out[tag + "c_high_fp16"] = a["c_high_fp16"].numpy()
                payload_arrays(tag, a["c_low_quant"], out)
                b = bases[pname][bkey]
                rec = ref_merge.reconstruct_from_coefficients(
                    a["c_high_fp16"].float(), quant.dequantize(a["c_low_quant"]).float(),
                    b["U_high"], b["U_low"], "cpu", mean=b["mean"])
                out[tag + "recon"] = rec.numpy()
    out["layout_json"] = np.array(json.dumps(layout, sort_keys=True))
    out["tasks"] = np.array(tasks)
    out["params"] = np.array(sorted(shapes))
    save("pipeline.npz", **out)


# ------------------------------------------------------------------------------- merge
def _merge_run():
    """merge_all_parameters + apply_merged_deltas (merge.py:304-552) over a toy model whose c_low always
    has >= 3 elements (6 tasks, max_rank 2), non-uniform weights, one masked parameter with a noise region."""
    shapes = {"a.weight": (96, 64), "a.bias": (96,), "b.weight": (80, 40)}
    tasks = ["T0", "T1", "T2", "T3", "T4", "T5"]
    cfg = types.SimpleNamespace(svd_low_bits=4, svd_rtvq_stages=2, svd_include_noise=True, svd_min_mask_size=10,
                                svd_energy_threshold=0.9, svd_max_rank=2, svd_center=True, svd_fp16=True,
                                svd_noise_shrink=0.5)
    weights = {"T0": 0.3, "T1": 0.1, "T2": 0.2, "T3": 0.15, "T4": 0.05, "T5": 0.2}
    g = torch.Generator().manual_seed(8)
    task_vectors = {t: {} for t in tasks}
    out = {}
    masks = {}
    for pi, (pname, shp) in enumerate(sorted(shapes.items())):
        ds = synthetic_deltas(int(np.prod(shp)), len(tasks), 300 + pi)
        for t, d in zip(tasks, ds):
            task_vectors[t][pname] = d.view(shp)
        out[f"in__{pname}"] = torch.stack(ds).numpy()
    masks["b.weight"] = torch.rand(shapes["b.weight"], generator=g) > 0.4
    out["mask__b.weight"] = masks["b.weight"].numpy()
    bases = {}
    for pname in sorted(shapes):
        mask = masks.get(pname)
        md, ud = [], []
        for t in tasks:
            delta = task_vectors[t][pname]
            if mask is not None and mask.shape == delta.shape:
                md.append(ref_masks.apply_mask_to_tensor(delta, mask))
                ud.append(ref_masks.get_unmasked_portion(delta, mask))
            else:
                md.append(delta.flatten())
        basis = ref_basis.construct_masked_basis(md, ud if ud else None, energy_threshold=cfg.svd_energy_threshold,
                                                 max_rank=cfg.svd_max_rank, center=cfg.svd_center, device="cpu",
                                                 include_noise=cfg.svd_include_noise)
        for region in ("masked", "noise"):
            if basis.get(region) is not None:
                basis[region]["U_high"] = basis[region]["U_high"].half()
                basis[region]["U_low"] = basis[region]["U_low"].half()
                assert basis[region]["N"] - basis[region]["k"] >= 3
        bases[pname] = basis
    compressed = ref_compress.compress_all_parameters(task_vectors, masks, bases, cfg, device="cpu")
    original_shapes = {n: torch.Size(s) for n, s in shapes.items()}
    merged = ref_merge.merge_all_parameters(compressed, bases, masks, weights, original_shapes, cfg, device="cpu",
                                            verbose=False)
    # diagnostics of the same run (diagnostics.py:72-321; Q1: no mean added back)
    cfg.svd_mask_strategy = "union"
    cfg.svd_weighting = "uniform"
    diag = ref_diag.compute_all_diagnostics(task_vectors, compressed, bases, masks, cfg, device="cpu")
    return dict(shapes=shapes, tasks=tasks, cfg=cfg, weights=weights, g=g, task_vectors=task_vectors, out=out,
                masks=masks, bases=bases, compressed=compressed, merged=merged, diag=diag)


def gen_merge():
    r = _merge_run()
    shapes, tasks, weights, g, task_vectors, out = r["shapes"], r["tasks"], r["weights"], r["g"], r["task_vectors"], r["out"]
    merged, diag = r["merged"], r["diag"]

    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (np.floating,)):
            return float(o)
        return o
    out["diagnostics_json"] = np.array(json.dumps(plain(diag), sort_keys=True))
    a, b = task_vectors["T0"]["a.weight"].flatten(), merged["a.weight"].flatten()
    em = ref_diag.compute_reconstruction_error(a, b)
    out["err_metrics_T0_vs_merged"] = np.array([em[k] for k in ("absolute_error", "relative_error",
                                                "max_absolute_error", "mean_absolute_error", "original_norm",
                                                "reconstructed_norm")], dtype=np.float64)
    base_sd = {n: torch.randn(s, generator=g) for n, s in shapes.items()}
    base_sd["extra.buffer"] = torch.arange(5, dtype=torch.float32)
    final = ref_merge.apply_merged_deltas(base_sd, merged, device="cpu", verbose=False)
    for n in shapes:
        assert torch.isfinite(merged[n]).all()
        out[f"merged__{n}"] = merged[n].numpy()
        out[f"base__{n}"] = base_sd[n].numpy()
        out[f"final__{n}"] = final[n].numpy()
    # weighted average of the exact task deltas: what the merge approximates
    for n in shapes:
        exact = sum(weights[t] * task_vectors[t][n] for t in tasks)
        out[f"exact__{n}"] = exact.numpy()
    out["tasks"] = np.array(tasks)
    out["weights"] = np.array([weights[t] for t in tasks], dtype=np.float64)
    out["params"] = np.array(sorted(shapes))
    save("merge.npz", **out)



def _diag_case(prefix, shapes, tasks, cfg, masks, seed0, out):
    """cli.py Step 4 + 5 + diagnostics of the reference on a toy model; stores, per parameter, what
    compute_parameter_diagnostics (diagnostics.py:120-231) read -- the masked originals, the (fp16-cast) basis, every task's
    c_high_fp16 and dequantized c_low -- and the six numbers it returned for every task."""
    task_vectors = {t: {} for t in tasks}
    for pi, (pname, shp) in enumerate(sorted(shapes.items())):
        for t, d in zip(tasks, synthetic_deltas(int(np.prod(shp)), len(tasks), seed0 + pi)):
            task_vectors[t][pname] = d.view(shp)
    bases = {}
    for pname in sorted(shapes):
        mask = masks.get(pname)
        md, ud = [], []
        for t in tasks:
            delta = task_vectors[t][pname]
            if mask is not None and mask.shape == delta.shape:
                md.append(ref_masks.apply_mask_to_tensor(delta, mask))
                ud.append(ref_masks.get_unmasked_portion(delta, mask))
            else:
                md.append(delta.flatten())
        basis = ref_basis.construct_masked_basis(md, ud if ud else None, energy_threshold=cfg.svd_energy_threshold,
                                                 max_rank=cfg.svd_max_rank, center=cfg.svd_center, device="cpu",
                                                 include_noise=cfg.svd_include_noise)
        if cfg.svd_fp16:
            for region in ("masked", "noise"):
                if basis.get(region) is not None:
                    basis[region]["U_high"] = basis[region]["U_high"].half()
                    basis[region]["U_low"] = basis[region]["U_low"].half()
        bases[pname] = basis
    compressed = ref_compress.compress_all_parameters(task_vectors, masks, bases, cfg, device="cpu")
    diag = ref_diag.compute_all_diagnostics(task_vectors, compressed, bases, masks, cfg, device="cpu")
    quant = ref_rtvq.RTVQQuantizer(cfg.svd_low_bits, cfg.svd_rtvq_stages)
    keys = ("absolute_error", "relative_error", "max_absolute_error", "mean_absolute_error", "original_norm",
            "reconstructed_norm")
    for pname in sorted(shapes):
        bm = bases[pname]["masked"]
        assert bm["N"] - bm["k"] >= 3, (pname, bm["k"])          # finite quantizer
        mask = masks.get(pname)
        xs, chs, cls, met = [], [], [], []
        for t in tasks:
            delta = task_vectors[t][pname]
            xs.append((ref_masks.apply_mask_to_tensor(delta, mask) if mask is not None else delta.flatten()).numpy())
            art = compressed[pname][t]["masked"]
            chs.append(art["c_high_fp16"].numpy())
            cls.append(quant.dequantize(art["c_low_quant"]).float().numpy().reshape(-1))
            e = diag["per_parameter"][pname]["reconstruction_errors"][t]
            met.append([float(e[k]) for k in keys])
        out[f"{prefix}__x__{pname}"] = np.stack(xs)
        out[f"{prefix}__U_high__{pname}"] = bm["U_high"].numpy()
        out[f"{prefix}__U_low__{pname}"] = bm["U_low"].numpy()
        out[f"{prefix}__c_high_fp16__{pname}"] = np.stack(chs)
        out[f"{prefix}__c_low_deq__{pname}"] = np.stack(cls)
        out[f"{prefix}__metrics__{pname}"] = np.array(met, dtype=np.float64)
        if bm["mean"] is not None:
            out[f"{prefix}__mean__{pname}"] = bm["mean"].squeeze(1).numpy()
        if mask is not None:
            out[f"{prefix}__mask__{pname}"] = mask.numpy()
    out[f"{prefix}__params"] = np.array(sorted(shapes))
    out[f"{prefix}__tasks"] = np.array(tasks)


def gen_diag():
    out = {}
    g = torch.Generator().manual_seed(21)
    # A: merge.npz's configuration (fp16 basis, centred: Q1 makes the error ~ ||mean|| / ||x||), one masked parameter
    shapes = {"a.weight": (96, 64), "a.bias": (96,), "b.weight": (80, 40)}
    cfg = types.SimpleNamespace(svd_low_bits=4, svd_rtvq_stages=2, svd_include_noise=False, svd_min_mask_size=10,
                                svd_energy_threshold=0.9, svd_max_rank=2, svd_center=True, svd_fp16=True,
                                svd_noise_shrink=0.5, svd_mask_strategy="union", svd_weighting="uniform")
    masks = {"b.weight": torch.rand(shapes["b.weight"], generator=g) > 0.4}
    _diag_case("A", shapes, [f"T{i}" for i in range(6)], cfg, masks, 300, out)
    # B: 20 tasks, fp32 basis, NOT centred (the error is then the small quantization / rounding error itself: the case in
    # which a wrong L1 / L-infinity or a dropped column shows), ragged sizes, one masked parameter
    shapes = {"w": (70, 73), "v": (257,), "e": (1, 3001)}
    cfg = types.SimpleNamespace(svd_low_bits=4, svd_rtvq_stages=2, svd_include_noise=False, svd_min_mask_size=10,
                                svd_energy_threshold=0.9, svd_max_rank=None, svd_center=False, svd_fp16=False,
                                svd_noise_shrink=0.5, svd_mask_strategy="union", svd_weighting="uniform")
    masks = {"w": torch.rand(shapes["w"], generator=g) > 0.3}
    _diag_case("B", shapes, [f"T{i:02d}" for i in range(20)], cfg, masks, 700, out)
    # C: 12 tasks, fp16 basis, not centred, 8-bit x 1 stage
    shapes = {"m": (33, 129), "s": (513,)}
    cfg = types.SimpleNamespace(svd_low_bits=8, svd_rtvq_stages=1, svd_include_noise=False, svd_min_mask_size=10,
                                svd_energy_threshold=0.9, svd_max_rank=None, svd_center=False, svd_fp16=True,
                                svd_noise_shrink=0.5, svd_mask_strategy="union", svd_weighting="uniform")
    _diag_case("C", shapes, [f"T{i:02d}" for i in range(12)], cfg, {}, 900, out)
    out["cases"] = np.array(["A", "B", "C"])
    save("diag.npz", **out)


# ------------------------------------------------------------------------------- artifact files (f3)
def _tree(o):
    """Structure of a saved object: key trees, tensor dtypes / shapes, python types (no values)."""
    if isinstance(o, torch.Tensor):
        return {"tensor": str(o.dtype).replace("torch.", ""), "shape": list(o.shape), "device": o.device.type}
    if isinstance(o, torch.Size):
        return {"torch.Size": list(o)}
    if isinstance(o, dict):
        return {"dict": {str(k): _tree(v) for k, v in o.items()}}
    if isinstance(o, (list, tuple)):
        return {type(o).__name__: [_tree(v) for v in o]}
    return type(o).__name__


def _json_tree(o):
    if isinstance(o, dict):
        return {k: _json_tree(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_json_tree(v) for v in o]
    return type(o).__name__


def artifact_manifest(root):
    """Everything that defines the on-disk format of an artifact directory: relative file names, and per file the
    key tree with dtypes and shapes (.pt, loaded weights-only) or key tree with value types (.json)."""
    man = {}
    for dirpath, _, files in os.walk(root):
        for f in sorted(files):
            full = os.path.join(dirpath, f)
            rel = os.path.relpath(full, root).replace(os.sep, "/")
            if f.endswith(".pt"):
                man[rel] = _tree(torch.load(full, map_location="cpu", weights_only=True))
            elif f.endswith(".json"):
                man[rel] = _json_tree(json.load(open(full)))
    return man


def gen_storage():
    """The reference's own writer (storage.py:52-338 save_all_artifacts, :392-409 save_merged_model) on the merge.npz
    run, with a parameter name that needs sanitising.  Stored: the in-memory inputs (tensors + plain python,
    weights-only loadable), the manifest of what the reference wrote, and the outcome of the reference's
    load_all_artifacts reading the files THIS package writes from the same inputs."""
    import tempfile
    from dataclasses import asdict, fields
    from src.svd_hybrid import storage as ref_storage
    from src.svd_hybrid import config as ref_config
    r = _merge_run()
    tasks = r["tasks"]
    rename = {"a.weight": "blk/0\\a.weight"}                 # "/" and "\\" -> "_" (storage.py:72)
    bases = {rename.get(n, n): b for n, b in r["bases"].items()}
    compressed = {rename.get(n, n): c for n, c in r["compressed"].items()}
    diag = dict(r["diag"])
    diag["per_parameter"] = {rename.get(n, n): v for n, v in diag["per_parameter"].items()}
    cfg = ref_config.SVDHybridConfig(tasks=tasks, svd_energy_threshold=0.9, svd_max_rank=2, svd_include_noise=True,
                                     checkpoint_dir="/ckpt", base_model_path="/ckpt/base.pt", device="cpu")
    with tempfile.TemporaryDirectory() as td:
        ref_dir = os.path.join(td, "ref")
        ref_storage.save_all_artifacts(bases, compressed, diag, cfg, ref_dir)
        ref_storage.save_merged_model(r["merged"], os.path.join(ref_dir, "out"))
        man_ref = artifact_manifest(ref_dir)
        # this package's writer on the same inputs, then the REFERENCE's reader on those files
        import svdq_amd.storage as our_storage
        import svdq_amd.config as our_config
        our_cfg = our_config.SVDHybridConfig(**asdict(cfg))
        our_dir = os.path.join(td, "ours")
        our_storage.save_all_artifacts(bases, compressed, diag, our_cfg, our_dir)
        our_storage.save_merged_model(r["merged"], os.path.join(our_dir, "out"))
        man_ours = artifact_manifest(our_dir)
        assert man_ours == man_ref, "writer mismatch"
        back = ref_storage.load_all_artifacts(our_dir, device="cpu")      # SVDHybridConfig(**json) must accept ours
        assert back["config"] == cfg
        assert sorted(back["bases"]) == sorted(bases) and sorted(back["compressed"]) == sorted(compressed)
        for n in bases:
            for region in ("masked", "noise"):
                if bases[n].get(region) is None:
                    continue
                for key in ("U_high", "U_low", "singular_values", "mean"):
                    assert torch.equal(back["bases"][n][region][key], bases[n][region][key].cpu())
            for t in tasks:
                a = back["compressed"][n][t]["masked"]
                assert torch.equal(a["c_high_fp16"], compressed[n][t]["masked"]["c_high_fp16"])
                for pa, pb in zip(a["c_low_quant"]["payloads"], compressed[n][t]["masked"]["c_low_quant"]["payloads"]):
                    assert torch.equal(pa["quantized"], pb["quantized"]) and torch.equal(pa["scale"], pb["scale"])
    meta = {"manifest": man_ref, "reference_load_all_artifacts_reads_our_files": True,
            "config_fields": [f.name for f in fields(ref_config.SVDHybridConfig)],
            "safe_names": {n: n.replace("/", "_").replace("\\", "_") for n in bases}}
    with open(os.path.join(HERE, "artifact_manifest.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    def plain(o):                      # numpy scalars (np.mean of the error lists) -> python: weights-only loadable
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)) and not isinstance(o, torch.Size):
            return [plain(v) for v in o]
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, np.floating):
            return float(o)
        return o
    torch.save({"bases": bases, "compressed": compressed, "diagnostics": plain(diag), "config": asdict(cfg),
                "merged": r["merged"]}, os.path.join(HERE, "artifact_inputs.pt"))
    chk = torch.load(os.path.join(HERE, "artifact_inputs.pt"), weights_only=True)      # must stay weights-only loadable
    assert sorted(chk) == ["bases", "compressed", "config", "diagnostics", "merged"]
    print(f"wrote artifact_manifest.json ({len(man_ref)} files), artifact_inputs.pt "
          f"({os.path.getsize(os.path.join(HERE, 'artifact_inputs.pt')) / 1024:.0f} KiB)")

# ------------------------------------------------------------------------------- ingest / TVQ
def gen_tvq():
    """TaskVector.__init__ / compute_task_vector (finetuned - base) and whole-tensor quantization through the
    reference's QuantizedFinetunedModel / QuantizedBaseAndTaskVector / QuantizedTaskVector."""
    g = torch.Generator().manual_seed(31)
    base = {"w1": torch.randn(40, 30, generator=g), "b1": torch.randn(37, generator=g),
            "w2": 0.5 * torch.randn(64, 33, generator=g), "big": torch.randn(9000, generator=g),
            "steps": torch.arange(4, dtype=torch.int64), "u8": torch.arange(6, dtype=torch.uint8)}
    tasks = ["A", "B", "C"]
    fts = {}
    for ti, t in enumerate(tasks):
        ft = {k: (v + 0.01 * (ti + 1) * torch.randn(v.shape, generator=g)) if v.dtype == torch.float32 else v.clone()
              for k, v in base.items()}
        fts[t] = ft
    fts["B"]["b1"] = torch.randn(38, generator=g)        # shape mismatch -> skipped for B
    del fts["C"]["w2"]                                      # missing -> skipped for C
    out = {"tasks": np.array(tasks), "keys": np.array(list(base.keys()))}
    for k, v in base.items():
        out[f"base__{k}"] = v.numpy()
    for t in tasks:
        for k, v in fts[t].items():
            out[f"ft__{t}__{k}"] = v.numpy()
        tv = ref_tv.TaskVector(base, fts[t], task_name=t, verbose=False)
        out[f"tv_keys__{t}"] = np.array(list(tv.vector.keys()))
        for k, v in tv.vector.items():
            out[f"tv__{t}__{k}"] = v.numpy()
        ctv = ref_loader.compute_task_vector(base, fts[t])
        out[f"ctv_keys__{t}"] = np.array(list(ctv.keys()))
        for k, v in ctv.items():
            if v.dtype == torch.float32:
                assert torch.equal(v, tv.vector[k])
    tvA = ref_tv.TaskVector(base, fts["A"], task_name="A", verbose=False)
    s = (tvA + ref_tv.TaskVector(base, fts["B"], task_name="B", verbose=False)) * 0.5
    out["sum_name"] = np.array(str(s.task_name))
    out["sum_keys"] = np.array(list(s.vector.keys()))
    for k, v in s.vector.items():
        out[f"sum__{k}"] = v.numpy()
    applied = tvA.apply_to(base, verbose=False)
    for k, v in applied.items():
        out[f"applied__{k}"] = v.numpy()

    def store_payloads(prefix, pay):
        out[f"{prefix}__keys"] = np.array(list(pay.keys()))
        for k, p in pay.items():
            out[f"{prefix}__q__{k}"] = p["quantized"].numpy()
            out[f"{prefix}__scale__{k}"] = p["scale"].numpy()
            if "zero_point" in p:
                out[f"{prefix}__zp__{k}"] = p["zero_point"].numpy()

    for method in ("asymmetric", "absmax"):
        for qbit in (8, 4, 3):
            tag = f"{method}{qbit}"
            qf = ref_tv.QuantizedFinetunedModel(fts["A"], qbit=qbit, method=method)
            store_payloads(f"qf__{tag}", qf.quantized_weights)
            for k, v in qf.dequantize().items():
                out[f"qf__{tag}__deq__{k}"] = v.numpy()
            for k, v in qf.get_task_vector(base).items():
                out[f"qf__{tag}__tv__{k}"] = v.numpy()
        qb = ref_tv.QuantizedBaseAndTaskVector(base, tvA, base_qbit=8, task_qbit=4, method=method)
        store_payloads(f"qb__{method}__base", qb.quantized_base)
        store_payloads(f"qb__{method}__task", qb.quantized_task)
        for k, v in qb.dequantize().items():
            out[f"qb__{method}__deq__{k}"] = v.numpy()
        qt = ref_tv.QuantizedTaskVector(qb.quantized_task, method=method)
        for k, v in qt.dequantize().items():
            out[f"qt__{method}__deq__{k}"] = v.numpy()
        for k, v in qt.apply_to(base).items():
            out[f"qt__{method}__applied__{k}"] = v.numpy()
    save("tvq.npz", **out)


# ------------------------------------------------------------------------------- clustering / weighting
def gen_cluster():
    """cluster_tasks (clustering.py:198-245), compute_cluster_statistics (:278-316), the compute_weights
    family (weighting.py:61-396) and merge_with_clustering (merge.py:555-626; inputs = merge.npz's)."""
    import tempfile
    out = {}
    tasks = ["Cars", "DTD", "EuroSAT", "GTSRB", "MNIST", "RESISC45", "SVHN"]
    shapes = {"a.weight": (96, 64), "a.bias": (96,), "b.weight": (80, 40)}
    group = [0, 0, 0, 1, 1, 2, 2]   # latent direction each task mostly follows
    g = torch.Generator().manual_seed(77)
    tv = {t: {} for t in tasks}
    for pname, shp in sorted(shapes.items()):
        n = int(np.prod(shp))
        dirs = torch.randn(3, n, generator=g)
        stack = []
        for ti, t in enumerate(tasks):
            d = 0.01 * (dirs[group[ti]] * (1.0 + 0.1 * ti) + 0.6 * torch.randn(n, generator=g))
            if not (pname == "a.bias" and t == "MNIST"):      # one task lacks one parameter (zeros row block)
                tv[t][pname] = d.view(shp)
            stack.append(d)
        out[f"in__{pname}"] = torch.stack(stack).numpy()
    out["tasks"] = np.array(tasks)
    out["params"] = np.array(sorted(shapes))
    out["missing"] = np.array(["MNIST", "a.bias"])
    for method in ("kmeans", "hierarchical"):
        for k in (2, 3):
            a = ref_cluster.cluster_tasks(tv, k, method=method)
            out[f"labels__{method}__k{k}"] = np.array([a[t] for t in tasks], dtype=np.int64)
    assign = ref_cluster.cluster_tasks(tv, 3, method="kmeans")
    st = ref_cluster.compute_cluster_statistics(tv, assign)
    cids = sorted(st.keys())
    out["stats__cids"] = np.array(cids, dtype=np.int64)
    out["stats__size"] = np.array([st[c]["size"] for c in cids], dtype=np.int64)
    for key in ("mean_distance_to_centroid", "max_distance_to_centroid", "min_distance_to_centroid"):
        out[f"stats__{key}"] = np.array([st[c][key] for c in cids], dtype=np.float64)
    # weights
    acc = {"cars": 0.85, "DTD": 0.72, "euro_sat": 0.93, "GTSRB": 0.99, "MNIST": 0.995, "resisc-45": 0.9}  # SVHN absent
    with tempfile.TemporaryDirectory() as td:
        pf = os.path.join(td, "acc.json")
        with open(pf, "w") as fh:
            json.dump(acc, fh)
        out["acc_json"] = np.array(json.dumps(acc))
        m = ref_weighting.load_performance_metrics(pf, tasks)
        out["metrics"] = np.array([m[t] for t in tasks], dtype=np.float64)
        for T in (0.1, 1.0, 5.0):
            w = ref_weighting.compute_weights(tasks, "performance", performance_file=pf, temperature=T)
            out[f"w_perf__T{T}"] = np.array([w[t] for t in tasks], dtype=np.float64)
    w = ref_weighting.compute_weights(tasks, "uniform")
    out["w_uniform"] = np.array([w[t] for t in tasks], dtype=np.float64)
    w = ref_weighting.compute_weights(tasks, "cluster", cluster_assignments=assign)
    out["w_cluster"] = np.array([w[t] for t in tasks], dtype=np.float64)
    cperf = {0: 0.9, 1: 0.5, 2: 0.7}
    w = ref_weighting.compute_weights(tasks, "cluster", cluster_assignments=assign, cluster_performance=cperf)
    out["w_cluster_perf"] = np.array([w[t] for t in tasks], dtype=np.float64)
    out["cluster_perf"] = np.array([cperf[c] for c in (0, 1, 2)], dtype=np.float64)
    ws = ref_weighting.get_weight_statistics(w)
    out["w_stats"] = np.array([ws[k] for k in ("min", "max", "mean", "std", "entropy")], dtype=np.float64)
    tens = {t: tv[t]["b.weight"] for t in tasks}
    out["weighted_avg__b.weight"] = ref_weighting.apply_weights_to_tensors(tens, w).numpy()
    mbc = ref_cluster.merge_by_cluster(tv, assign, w)
    final = ref_cluster.merge_cluster_results(mbc, cperf)
    for pname in shapes:
        out[f"mbc_final__{pname}"] = final[pname].numpy()

    # merge_with_clustering on the merge.npz setting (same seeds as gen_merge)
    mshapes = {"a.weight": (96, 64), "a.bias": (96,), "b.weight": (80, 40)}
    mtasks = ["T0", "T1", "T2", "T3", "T4", "T5"]
    cfg = types.SimpleNamespace(svd_low_bits=4, svd_rtvq_stages=2, svd_include_noise=True, svd_min_mask_size=10,
                                svd_energy_threshold=0.9, svd_max_rank=2, svd_center=True, svd_fp16=True,
                                svd_noise_shrink=0.5)
    weights = {"T0": 0.3, "T1": 0.1, "T2": 0.2, "T3": 0.15, "T4": 0.05, "T5": 0.2}
    gm = torch.Generator().manual_seed(8)
    mtv = {t: {} for t in mtasks}
    for pi, (pname, shp) in enumerate(sorted(mshapes.items())):
        ds = synthetic_deltas(int(np.prod(shp)), len(mtasks), 300 + pi)
        for t, d in zip(mtasks, ds):
            mtv[t][pname] = d.view(shp)
    masks = {"b.weight": torch.rand(mshapes["b.weight"], generator=gm) > 0.4}
    bases = {}
    for pname in sorted(mshapes):
        mask = masks.get(pname)
        md, ud = [], []
        for t in mtasks:
            delta = mtv[t][pname]
            if mask is not None:
                md.append(ref_masks.apply_mask_to_tensor(delta, mask))
                ud.append(ref_masks.get_unmasked_portion(delta, mask))
            else:
                md.append(delta.flatten())
        basis = ref_basis.construct_masked_basis(md, ud if ud else None, energy_threshold=cfg.svd_energy_threshold,
                                                 max_rank=cfg.svd_max_rank, center=cfg.svd_center, device="cpu",
                                                 include_noise=cfg.svd_include_noise)
        for region in ("masked", "noise"):
            if basis.get(region) is not None:
                basis[region]["U_high"] = basis[region]["U_high"].half()
                basis[region]["U_low"] = basis[region]["U_low"].half()
        bases[pname] = basis
    compressed = ref_compress.compress_all_parameters(mtv, masks, bases, cfg, device="cpu")
    original_shapes = {n: torch.Size(sh) for n, sh in mshapes.items()}
    massign = {"T0": 0, "T1": 0, "T2": 1, "T3": 1, "T4": 1, "T5": 2}
    merged = ref_merge.merge_with_clustering(compressed, bases, masks, weights, massign, original_shapes, cfg,
                                             device="cpu")
    cstats = ref_diag.compute_compression_statistics(mtv, compressed, bases, cfg)
    out["compression_stats_json"] = np.array(json.dumps(cstats, sort_keys=True))
    out["mwc__assign"] = np.array([massign[t] for t in mtasks], dtype=np.int64)
    for pname in mshapes:
        assert torch.isfinite(merged[pname]).all()
        out[f"mwc__{pname}"] = merged[pname].numpy()
    save("cluster.npz", **out)



# ------------------------------------------------------------------------------- graded spectra
def _structured(D, N, sigmas, seed, common_mean=0.0):
    """T = Q diag(sigmas) Z^T (Q: D x N and Z: N x N orthonormal, built in fp64) cast to fp32, plus an optional
    common per-row offset so that centring has something to remove."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, N)))
    Z, _ = np.linalg.qr(rng.standard_normal((N, N)))
    T = (Q * np.asarray(sigmas, dtype=np.float64)) @ Z.T
    if common_mean:
        T = T + common_mean * rng.standard_normal((D, 1))
    return [torch.from_numpy(np.ascontiguousarray(T[:, j]).astype(np.float32)) for j in range(N)]


def _structured_centred(D, N, sigmas, seed, common_mean):
    """T = Q diag(sigmas) Z^T + c 1^T with the N - 1 columns of Z orthonormal AND orthogonal to the ones vector, so that
    the CENTRED stack has exactly the singular values given (N - 1 of them) and the common offset c is what centring
    removes."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, N - 1)))
    M = np.concatenate([np.ones((N, 1)), rng.standard_normal((N, N - 1))], axis=1)
    Zf, _ = np.linalg.qr(M)
    Z = Zf[:, 1:]                      # orthonormal complement of 1 / sqrt(N)
    T = (Q * np.asarray(sigmas, dtype=np.float64)) @ Z.T + common_mean * rng.standard_normal((D, 1))
    return [torch.from_numpy(np.ascontiguousarray(T[:, j]).astype(np.float32)) for j in range(N)]


def _s64(deltas, center):
    """fp64 singular values of exactly the matrix the reference factorises (fp32 stack, fp32 centring)."""
    T = torch.stack(deltas, dim=1)
    if center:
        T = T - T.mean(dim=1, keepdim=True)
    return np.linalg.svd(T.double().numpy(), compute_uv=False)


def gen_spectrum():
    """Chains whose spectrum reaches far below sigma_0: graded singular values, exactly dependent
    tasks, and a cumulative energy that sits 5e-5 below / above the threshold."""
    graded8 = [1.0, 1e-1, 1e-2, 1e-3, 1e-4, 3e-5, 1e-5, 3e-6]
    graded16 = [10.0 ** (-i / 3.0) for i in range(16)]
    specs = []
    specs.append(("spectrum_graded_n8", _structured(6000, 8, graded8, 101), 0.99999, False))
    specs.append(("spectrum_graded_n8c", _structured(6000, 8, graded8, 102, common_mean=0.3), 0.99999, True))
    specs.append(("spectrum_graded_n16", _structured(5000, 16, graded16, 103), 0.99995, False))
    # N > 16: fp32-product Gram first, then the flagged fp64 pass
    specs.append(("spectrum_graded_n20", _structured(5000, 20, [10.0 ** (-i / 4.0) for i in range(20)], 107),
                  0.99995, False))
    # exactly dependent tasks: t2 = t0, t4 = t0 + t1 -> rank 4 of 6
    base = _structured(5000, 4, [1.0, 0.5, 0.25, 0.125], 104)
    dep = [base[0], base[1], base[0].clone(), base[2], base[0] + base[1], base[3]]
    specs.append(("spectrum_rankdef_n6", dep, 0.9, False))
    # two identical tasks, centred (VERDICT: "a rank-deficient input (two identical tasks)")
    b5 = _structured(5000, 5, [1.0, 0.6, 0.4, 0.3, 0.2], 105, common_mean=0.2)
    specs.append(("spectrum_twins_n6c", [b5[0], b5[1], b5[2], b5[1].clone(), b5[3], b5[4]], 0.9, True))
    # cumulative energy of the first two directions = 0.9 -/+ 5e-5
    for tag, e1 in (("below", 0.29995), ("above", 0.30005)):
        en = [0.6, e1, 0.05 + (0.3 - e1), 0.03, 0.01, 0.006, 0.003, 0.001]
        specs.append((f"spectrum_thresh_{tag}_n8", _structured(6000, 8, np.sqrt(en), 106), 0.9, False))
    _save_spectrum_specs(specs)


def _save_spectrum_specs(specs):
    for name, deltas, thr, center in specs:
        out = basis_chain(deltas, thr, None, center, True, 4, 2, store_inputs=True, store_full=True)
        out["S_f64"] = _s64(deltas, center)
        save(name + ".npz", **out)
        print("   S  ", out["S"], " k", int(out["k"]), " energy", float(out["energy_retained"]))


def gen_spectrum_gap():
    """N > 16 with ONE singular value near 1e-4 sigma_0 behind a gap: lambda = 1e-8 lambda_0 lies below
    what fp32-product sums resolve, so the first-pass Gram may measure it as negative (clipped to sigma = 0) -- the
    refinement trigger must fire on the surplus null direction, not only on the (3e-7, 2e-2) sigma_0 band.  Two
    seeds, centred and not, so that both signs of the first-pass noise are likely to occur."""
    tail = [1.0, 0.8, 0.65, 0.5, 0.4, 0.32, 0.25, 0.2, 0.16, 0.13, 0.1, 0.085, 0.07, 0.06, 0.05, 0.045, 0.04, 0.035, 0.03]
    specs = [("spectrum_gap_n20a", _structured(5000, 20, tail + [1.0e-4], 108), 0.99, False),
             ("spectrum_gap_n20b", _structured_centred(5000, 20, tail[:18] + [1.3e-4], 109, 0.2), 0.99, True)]
    _save_spectrum_specs(specs)

# ------------------------------------------------------------------------------- rank KATs
def gen_rank():
    out = {}
    spectra = {
        "decay8": [10.0, 5.0, 2.0, 1.0, 0.5, 0.2, 0.1, 0.05],
        "tiny": [10.0, 1e-10, 1e-12],
        "dominant": [100.0, 0.01, 0.001],
        "four": [4.0, 3.0, 2.0, 1.0],
        "five": [10.0, 5.0, 2.0, 1.0, 0.5],
        "zeros": [0.0, 0.0, 0.0],
        "ones100": [1.0] * 100,
    }
    for name, vals in spectra.items():
        S = torch.tensor(vals)
        out[f"{name}__S"] = S.numpy()
        out[f"{name}__cum"] = ref_basis.compute_energy_spectrum(S).numpy()
        for thr in (0.5, 0.9, 0.95, 0.99, 0.999, 1.0):
            for mr in (None, 2, 10):
                key = f"{name}__k__thr{thr}__mr{mr}"
                out[key] = np.int64(ref_basis.select_rank(S, thr, mr))
        out[f"{name}__k__minrank2"] = np.int64(ref_basis.select_rank(S, 0.999, None, min_rank=2))
    save("rank_kats.npz", **out)


def gen_api():
    """api_signatures.json: the public callables of the reference's in-scope modules -- names, argument names and
    defaults as text (inspect.signature of the imported reference; class methods as Class.method).  Data about the
    interface, not source: tests/test_abi_cpu.py checks that this package offers every one of them with the same
    leading arguments."""
    import inspect
    import task_vectors as ref_tv
    from src.svd_hybrid import storage as ref_storage, task_vector_loader as ref_tvl, reload as ref_reload

    def sig(fn):
        out = []
        for name, prm in inspect.signature(fn).parameters.items():
            if prm.kind in (prm.VAR_POSITIONAL, prm.VAR_KEYWORD):
                out.append(("*" if prm.kind == prm.VAR_POSITIONAL else "**") + name)
            else:
                out.append(name if prm.default is prm.empty else f"{name}={prm.default!r}")
        return out

    mods = {"basis": ref_basis, "compress": ref_compress, "rtvq": ref_rtvq, "mask_loader": ref_masks, "merge": ref_merge,
            "diagnostics": ref_diag, "storage": ref_storage, "weighting": ref_weighting, "clustering": ref_cluster,
            "task_vector_loader": ref_tvl, "reload": ref_reload, "quantization_utils": ref_qutils,
            "task_vectors": ref_tv}
    api = {}
    for mname, mod in mods.items():
        entry = {}
        for name, obj in vars(mod).items():
            if name.startswith("_") or getattr(obj, "__module__", None) != mod.__name__ or name == "main":
                continue
            if inspect.isfunction(obj):
                entry[name] = sig(obj)
            elif inspect.isclass(obj):
                entry[name] = sig(obj.__init__)[1:]
                for meth, mobj in vars(obj).items():
                    if inspect.isfunction(mobj) and not meth.startswith("_"):
                        entry[f"{name}.{meth}"] = sig(mobj)[1:]
        api[mname] = entry
    with open(os.path.join(HERE, "api_signatures.json"), "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)
    print("api_signatures.json:", {m: len(v) for m, v in api.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate only the named families, e.g. `make_golden.py cluster`
        for fam in sys.argv[1:]:
            globals()["gen_" + fam]()
        sys.exit(0)
    gen_rtvq()
    gen_rtvq_large()
    gen_rank()
    gen_basis()
    gen_spectrum()
    gen_spectrum_gap()
    gen_config1()
    gen_masks()
    gen_pipeline()
    gen_merge()
    gen_diag()
    gen_storage()
    gen_cluster()
    gen_tvq()
    gen_api()
