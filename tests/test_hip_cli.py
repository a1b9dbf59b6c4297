"""
End-to-end test of the reference-shaped driver (svdq_amd.cli, scripts/run_svd_hybrid.py, scripts/reload_svd_hybrid.py,
scripts/load_and_merge.py)
on synthetic checkpoints written to disk: flags -> config -> the whole pipeline on the GPU -> the reference's output
files -> reload from artifacts.  Also pins compute_compression_statistics to the reference's numbers.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sq():
    import svdq_amd
    return svdq_amd


def _write_checkpoints(tmp, tasks, with_masks):
    from oracle.svd_hybrid_oracle import synthetic_deltas
    g = torch.Generator().manual_seed(123)
    shapes = {"blk.attn.weight": (96, 64), "blk.attn.bias": (96,), "blk.mlp.weight": (128, 64), "ln.weight": (64,)}
    base = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    base["steps"] = torch.tensor(7)                                    # integer buffer: carried over untouched
    ck = tmp / "ckpt"
    ck.mkdir()
    torch.save({"state_dict": base}, tmp / "base.pt")                  # wrapped form (task_vector_loader.py:88-98)
    deltas = {k: synthetic_deltas(int(np.prod(s)), len(tasks), 700 + i) for i, (k, s) in enumerate(shapes.items())}
    for ti, t in enumerate(tasks):
        sd = {k: base[k] + deltas[k][ti].view(shapes[k]) for k in shapes}
        sd["steps"] = torch.tensor(9)
        if ti % 2:
            torch.save(sd, ck / f"{t}.pt")
        else:
            (ck / t).mkdir()
            torch.save({"model_state_dict": sd}, ck / t / "finetuned.pt")
    if with_masks:
        md = tmp / "masks"
        md.mkdir()
        for t in tasks:
            torch.save({"blk.mlp.weight": torch.rand(shapes["blk.mlp.weight"], generator=g) > 0.6}, md / f"{t}_mask.pt")
    return base, shapes, deltas


@pytest.mark.parametrize("weighting,with_masks", [("uniform", False), ("cluster", True), ("performance", False)])
def test_cli_end_to_end(sq, tmp_path, weighting, with_masks):
    tasks = ["Cars", "DTD", "EuroSAT", "GTSRB", "MNIST", "SVHN"]
    base, shapes, deltas = _write_checkpoints(tmp_path, tasks, with_masks)
    out, art = tmp_path / "out", tmp_path / "art"
    argv = ["--tasks", *tasks, "--checkpoint-dir", str(tmp_path / "ckpt"), "--base-model-path", str(tmp_path / "base.pt"),
            "--energy-threshold", "0.9", "--max-rank", "2", "--low-bits", "4", "--rtvq-stages", "2",
            "--weighting", weighting, "--cluster-k", "2", "--store-artifacts", "--output-dir", str(out),
            "--artifact-dir", str(art)]
    if with_masks:
        argv += ["--mask-dir", str(tmp_path / "masks"), "--include-noise"]
    if weighting == "performance":
        pf = tmp_path / "acc.json"
        pf.write_text(json.dumps({t.lower(): 0.5 + 0.05 * i for i, t in enumerate(tasks)}))
        argv += ["--performance-file", str(pf), "--temperature", "0.5"]
    res = sq.cli.main(argv)
    merged = torch.load(out / "merged_state_dict.pt", map_location="cpu", weights_only=True)
    weights = json.loads((out / "weights.json").read_text())
    assert abs(sum(weights.values()) - 1.0) < 1e-6 and sorted(weights) == sorted(tasks)
    if weighting == "cluster":
        assign = json.loads((out / "clusters.json").read_text())
        assert sorted(assign) == sorted(tasks) and len(set(assign.values())) == 2
    if weighting == "performance":
        assert weights["SVHN"] > weights["Cars"]
    assert set(merged.keys()) == set(base.keys()) and int(merged["steps"]) == 7
    for k, s in shapes.items():
        exact = sum(weights[t] * deltas[k][i].view(s) for i, t in enumerate(tasks))
        got = merged[k] - base[k]
        assert got.shape == torch.Size(s)
        if weighting != "cluster" and not with_masks:        # the merged delta approximates the weighted mean
            rel = float((got - exact).norm() / exact.norm())
            assert rel < 0.2, (k, rel)
    # the reference's artifact layout
    for sub in ("basis", "coeffs"):
        assert (art / sub).is_dir() and len(list((art / sub).iterdir())) == len(shapes)
    diag = json.loads((art / "diagnostics.json").read_text())
    assert diag["task_weights"] == weights and "summary" in diag
    cfgj = json.loads((art / "config.json").read_text())
    assert cfgj["svd_energy_threshold"] == 0.9 and cfgj["svd_max_rank"] == 2 and cfgj["svd_weighting"] == weighting
    st = res["compression_statistics"]["summary"]
    assert st["num_parameters"] == len(shapes) and st["num_tasks"] == len(tasks) and st["overall_compression_ratio"] > 1
    if not with_masks and weighting == "uniform":
        # reload script: merged model rebuilt from artifacts + base only
        rc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "reload_svd_hybrid.py"), "--artifact-dir",
                             str(art), "--base-model-path", str(tmp_path / "base.pt"), "--verify",
                             str(out / "merged_state_dict.pt")], capture_output=True, text=True, timeout=300)
        assert rc.returncode == 0 and "MATCH" in rc.stdout, rc.stdout + rc.stderr
        # the reference's top-level load_and_merge.py: the same from its four flags (default --device cpu: results on
        # the host), written to --output-path
        outp = tmp_path / "merged_again.pt"
        rc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "load_and_merge.py"), "--artifact-dir",
                             str(art), "--base-model-path", str(tmp_path / "base.pt"), "--output-path", str(outp)],
                            capture_output=True, text=True, timeout=300)
        assert rc.returncode == 0, rc.stdout + rc.stderr
        again = torch.load(outp, map_location="cpu", weights_only=True)
        saved = torch.load(out / "merged_state_dict.pt", map_location="cpu", weights_only=True)
        assert set(again) == set(saved)
        for key, v in saved.items():
            assert torch.allclose(again[key].float(), v.float(), atol=1e-5), key


def test_cli_argument_errors(sq):
    with pytest.raises(ValueError, match="--tasks must be specified"):
        sq.cli.main(["--checkpoint-dir", "x", "--base-model-path", "y"])
    with pytest.raises(ValueError, match="--checkpoint-dir must be specified"):
        sq.cli.main(["--tasks", "A", "--base-model-path", "y"])
    with pytest.raises(FileNotFoundError):
        sq.cli.main(["--tasks", "A", "--checkpoint-dir", "/nonexistent", "--base-model-path", "/nonexistent/base.pt"])


def test_compression_statistics_vs_reference(sq):
    g, gc = load_golden("merge.npz"), load_golden("cluster.npz")
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
    got = sq.compute_compression_statistics(tv, comp, bases, cfg)
    ref = json.loads(str(gc["compression_stats_json"]))
    assert json.loads(json.dumps(got, sort_keys=True)) == ref


@pytest.mark.parametrize("strategy", ["union", "majority"])
def test_packed_tall_mask_file_route(sq, tmp_path, strategy):
    """A TALL_mask_{N}task file (bit-packed per-task masks over the flattened, key-sorted state dict): the packed
    GPU route equals load_tall_mask_file + combine_masks, and the command line picks it up."""
    tasks = ["Cars", "DTD", "EuroSAT", "GTSRB", "MNIST", "SVHN"]
    base, shapes, deltas = _write_checkpoints(tmp_path, tasks, with_masks=False)
    keys = sorted(base.keys())
    total = sum(base[k].numel() for k in keys)
    rng = np.random.default_rng(3)
    packed = {t: np.packbits(rng.random(total) > 0.55) for t in tasks}
    md = tmp_path / "tall"
    md.mkdir()
    np.savez(md / f"TALL_mask_{len(tasks)}task.npz", **packed)
    path = str(md / f"TALL_mask_{len(tasks)}task.npz")
    per_task = sq.load_tall_mask_file(path, base, device="cuda")
    want = sq.combine_masks({t: per_task[t] for t in tasks}, strategy=strategy, device="cuda", verbose=False)
    got = sq.combine_tall_masks_packed(path, tasks, base, strategy, "cuda")
    assert sorted(got) == sorted(want) == keys
    for k in keys:
        assert got[k].shape == base[k].shape and got[k].dtype == torch.bool
        assert torch.equal(got[k], want[k].to(got[k].device)), k
    # a task the file does not contain is skipped (combine_masks skips tasks whose masks are None)
    got5 = sq.combine_tall_masks_packed(path, tasks[:5] + ["Unknown"], base, strategy, "cuda")
    want5 = sq.combine_masks({t: per_task[t] for t in tasks[:5]}, strategy=strategy, device="cuda", verbose=False)
    assert all(torch.equal(got5[k], want5[k].to(got5[k].device)) for k in keys)
    out, art = tmp_path / "out", tmp_path / "art"
    res = sq.cli.main(["--tasks", *tasks, "--checkpoint-dir", str(tmp_path / "ckpt"), "--base-model-path",
                       str(tmp_path / "base.pt"), "--mask-dir", str(md), "--mask-strategy", strategy,
                       "--energy-threshold", "0.9", "--max-rank", "2", "--output-dir", str(out), "--artifact-dir", str(art)])
    for k, s in shapes.items():
        b = res["bases"][k]["masked"]
        assert b["D"] == int(want[k].sum()) and b["U_high"].shape[0] == b["D"]


@pytest.mark.parametrize("with_masks", [False, True])
def test_cli_from_checkpoints_route_matches(sq, tmp_path, with_masks):
    """--no-eval-reconstruction without clustering: the driver compresses straight from the fine-tuned and base weights
    (no task vectors in HBM; masked parameters through gather + minus-base in one pass); the merged model is the same,
    bit for bit, as on the ordinary route."""
    tasks = ["Cars", "DTD", "EuroSAT", "GTSRB", "MNIST", "SVHN"]
    base, shapes, deltas = _write_checkpoints(tmp_path, tasks, with_masks=with_masks)
    common = ["--tasks", *tasks, "--checkpoint-dir", str(tmp_path / "ckpt"), "--base-model-path", str(tmp_path / "base.pt"),
              "--energy-threshold", "0.9", "--max-rank", "2", "--store-artifacts"]
    if with_masks:
        common += ["--mask-dir", str(tmp_path / "masks"), "--include-noise"]
    a = sq.cli.main(common + ["--output-dir", str(tmp_path / "o1"), "--artifact-dir", str(tmp_path / "a1")])
    b = sq.cli.main(common + ["--no-eval-reconstruction", "--output-dir", str(tmp_path / "o2"), "--artifact-dir",
                              str(tmp_path / "a2")])
    assert "summary" in a["diagnostics"] and list(b["diagnostics"].keys()) == ["task_weights"]
    assert set(a["merged_state_dict"]) == set(b["merged_state_dict"])
    for k in a["merged_state_dict"]:
        assert torch.equal(a["merged_state_dict"][k].cpu(), b["merged_state_dict"][k].cpu()), k
    for k in shapes:
        assert torch.equal(a["bases"][k]["masked"]["U_low"], b["bases"][k]["masked"]["U_low"])
    assert a["compression_statistics"]["summary"] == b["compression_statistics"]["summary"]


def test_main_dispatcher_runs_svd_hybrid(sq, tmp_path):
    """reference src/main.py:41-89: `--method svd_hybrid` (also the default) hands the remaining flags to the SVD-Hybrid
    command line; as a subprocess, like a user would call it."""
    tasks = ["Cars", "DTD", "EuroSAT", "GTSRB"]
    base, shapes, deltas = _write_checkpoints(tmp_path, tasks, False)
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "main.py"), "--method", "svd_hybrid", "--tasks", *tasks,
           "--checkpoint-dir", str(tmp_path / "ckpt"), "--base-model-path", str(tmp_path / "base.pt"),
           "--energy-threshold", "0.9", "--max-rank", "1", "--output-dir", str(out), "--artifact-dir", str(tmp_path / "art")]
    rc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert rc.returncode == 0, rc.stdout[-2000:] + rc.stderr[-2000:]
    merged = torch.load(out / "merged_state_dict.pt", map_location="cpu", weights_only=True)
    assert set(merged.keys()) == set(base.keys())
