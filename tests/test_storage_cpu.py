"""
Artifact format: files, keys, dtypes and round trip.  Host code only,
so this runs without a GPU on hand-made dicts shaped like the hot path's outputs.
"""
import json
import os
import sys
from dataclasses import fields

import torch

import svdq_amd as sq

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import GOLDEN, artifact_manifest  # noqa: E402


def _basis(D, N, k, mean=True):
    return {"U_high": torch.randn(D, k).half(), "U_low": torch.randn(D, N - k).half(),
            "singular_values": torch.rand(N), "k": k, "mean": torch.randn(D, 1) if mean else None,
            "energy_retained": 0.93, "D": D, "N": N, "_svdq_batch": object()}


def _quant(n, bits=4, stages=2):
    return {"payloads": [{"stage": s, "quantized": torch.randint(0, 16, (n,), dtype=torch.uint8),
                          "scale": torch.tensor(3.5), "zero_point": torch.tensor(7.0), "residual_norm": 0.25}
                         for s in range(stages)],
            "num_bits": bits, "num_stages": stages, "original_shape": torch.Size([n]), "original_dtype": "torch.float32"}


def test_artifact_layout_and_round_trip(tmp_path):
    tasks = ["Cars", "DTD", "EuroSAT"]
    bases = {"blk/0.attn.weight": {"masked": _basis(40, 3, 1), "noise": _basis(12, 3, 2)},
             "blk.0.attn.bias": {"masked": _basis(8, 3, 1, mean=False), "noise": None}}
    comp = {n: {t: {"masked": {"c_high_fp16": torch.randn(b["masked"]["k"]).half(),
                               "c_low_quant": _quant(3 - b["masked"]["k"])},
                    "unmasked": ({"c_high_fp16": torch.randn(b["noise"]["k"]).half(),
                                  "c_low_quant": _quant(3 - b["noise"]["k"])} if b["noise"] else None)}
                for t in tasks} for n, b in bases.items()}
    diag = {"config": {"svd_low_bits": 4}, "per_parameter": {n: {"param_name": n, "original_shape": torch.Size([4, 10]),
                                                                  "masked_size": torch.tensor(40)} for n in bases},
            "summary": {"num_parameters": 2}, "task_weights": {t: 1 / 3 for t in tasks}}
    cfg = sq.SVDHybridConfig(tasks=tasks, svd_include_noise=True)
    d = str(tmp_path / "artifacts")
    sq.save_all_artifacts(bases, comp, diag, cfg, d)
    assert sorted(os.listdir(d)) == ["basis", "coeffs", "config.json", "diagnostics.json"]
    assert sorted(os.listdir(os.path.join(d, "basis"))) == ["blk.0.attn.bias.pt", "blk_0.attn.weight.pt"]
    assert sorted(os.listdir(os.path.join(d, "coeffs"))) == ["blk.0.attn.bias.pt", "blk_0.attn.weight.pt"]
    raw = torch.load(os.path.join(d, "basis", "blk_0.attn.weight.pt"), weights_only=True)
    assert sorted(raw.keys()) == ["masked", "noise"]
    assert sorted(raw["masked"].keys()) == sorted(["U_high", "U_low", "singular_values", "k", "mean", "energy_retained", "D", "N"])
    assert raw["masked"]["U_high"].dtype == torch.float16 and raw["masked"]["mean"].shape == (40, 1)
    assert "noise" not in torch.load(os.path.join(d, "basis", "blk.0.attn.bias.pt"), weights_only=True)
    cj = json.load(open(os.path.join(d, "config.json")))
    assert cj["svd_energy_threshold"] == 0.95 and cj["tasks"] == tasks
    dj = json.load(open(os.path.join(d, "diagnostics.json")))
    assert dj["per_parameter"]["blk/0.attn.weight"]["original_shape"] == [4, 10]
    assert dj["per_parameter"]["blk/0.attn.weight"]["masked_size"] == 40
    art = sq.load_all_artifacts(d)
    assert sorted(art.keys()) == ["bases", "compressed", "config", "diagnostics"]
    assert art["config"] == cfg
    for n in bases:
        for region in ("masked", "noise"):
            if bases[n][region] is None:
                assert region not in art["bases"][n]
                continue
            for key in ("U_high", "U_low", "singular_values"):
                assert torch.equal(art["bases"][n][region][key], bases[n][region][key])
            assert art["bases"][n][region]["k"] == bases[n][region]["k"]
        for t in tasks:
            got = art["compressed"][n][t]["masked"]
            assert torch.equal(got["c_high_fp16"], comp[n][t]["masked"]["c_high_fp16"])
            assert got["c_low_quant"]["original_shape"] == comp[n][t]["masked"]["c_low_quant"]["original_shape"]
            assert (comp[n][t]["unmasked"] is None) == ("unmasked" not in art["compressed"][n][t])
    for fn in (lambda: sq.load_basis("nope", d), lambda: sq.load_compressed_coefficients("nope", d),
               lambda: sq.load_config(str(tmp_path)), lambda: sq.load_diagnostics(str(tmp_path))):
        try:
            fn()
            assert False
        except FileNotFoundError:
            pass
    sq.save_merged_model({"w": torch.ones(2)}, str(tmp_path / "out"))
    assert os.path.exists(tmp_path / "out" / "merged_state_dict.pt")


def test_writer_reproduces_the_reference_writers_files(tmp_path):
    """f3 pinned to the reference writer: tests/golden/make_golden.py::gen_storage ran the reference's
    save_all_artifacts / save_merged_model (storage.py:52-338, :392-409) on a 3-parameter x 6-task run (one name that
    needs sanitising, one masked parameter with a noise basis) and recorded the manifest of what it wrote -- file
    names, key trees, dtypes, shapes, json value types -- plus the in-memory inputs.  This package's writer, fed the
    same inputs, must produce the same manifest.  (At generation time the reference's load_all_artifacts also read
    the files this package wrote; that outcome is recorded in the manifest file.)"""
    meta = json.load(open(os.path.join(GOLDEN, "artifact_manifest.json")))
    assert meta["reference_load_all_artifacts_reads_our_files"] is True
    inp = torch.load(os.path.join(GOLDEN, "artifact_inputs.pt"), map_location="cpu", weights_only=True)
    cfg = sq.SVDHybridConfig(**inp["config"])
    d = str(tmp_path / "ours")
    sq.save_all_artifacts(inp["bases"], inp["compressed"], inp["diagnostics"], cfg, d)
    sq.save_merged_model(inp["merged"], os.path.join(d, "out"))
    assert artifact_manifest(d) == meta["manifest"]
    for name, safe in meta["safe_names"].items():
        assert os.path.exists(os.path.join(d, "basis", safe + ".pt")), (name, safe)
    # config.json carries exactly the reference dataclass's fields (its load_config does SVDHybridConfig(**json))
    cj = json.load(open(os.path.join(d, "config.json")))
    assert sorted(cj) == sorted(meta["config_fields"])
    ours = {f.name for f in fields(sq.SVDHybridConfig)}
    assert ours - set(meta["config_fields"]) == {"svd_low_bits_by_param"}      # the one extension, never written
    # and the weights-only readers take everything back
    art = sq.load_all_artifacts(d)
    assert sorted(art["bases"]) == sorted(inp["bases"]) and art["config"] == cfg
    for n, b in inp["bases"].items():
        for region in ("masked", "noise"):
            if b.get(region) is not None:
                assert torch.equal(art["bases"][n][region]["U_high"], b[region]["U_high"])
        for t, a in inp["compressed"][n].items():
            got = art["compressed"][n][t]["masked"]["c_low_quant"]
            assert got["original_shape"] == a["masked"]["c_low_quant"]["original_shape"]
            assert torch.equal(got["payloads"][0]["quantized"], a["masked"]["c_low_quant"]["payloads"][0]["quantized"])


def test_mixed_width_config_is_not_written(tmp_path):
    cfg = sq.SVDHybridConfig(tasks=["a", "b"], svd_low_bits_by_param=lambda n: 8 if n.endswith("weight") else 2)
    sq.storage.save_config(cfg, str(tmp_path))
    cj = json.load(open(tmp_path / "config.json"))
    assert "svd_low_bits_by_param" not in cj
    assert sq.load_config(str(tmp_path)).svd_low_bits == cfg.svd_low_bits
