"""
CPU-only checks of the drop-in boundary: the C ABI library builds for gfx950, loads, and exports
exactly the symbols include/svdq.h declares (no compute calls without a GPU); the Python host
layer mirrors the reference's names/defaults/errors; a missing GPU or library fails loudly.
"""
import inspect
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sq():
    import svdq_amd
    if not os.path.exists(svdq_amd._native.LIB_PATH):
        svdq_amd._native.build()
    return svdq_amd


def _header_functions():
    text = open(os.path.join(ROOT, "include", "svdq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svdq_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_exports_and_binding(sq):
    declared = _header_functions()
    out = subprocess.check_output(["nm", "-D", "--defined-only", sq._native.LIB_PATH], text=True)
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if " T svdq_" in ln})
    assert declared == exported, (set(declared) ^ set(exported))
    assert sorted(sq._native.SIGNATURES.keys()) == declared
    lib = sq._native.lib()
    assert lib.svdq_abi_version() == 1
    for name in declared:
        assert hasattr(lib, name)


def test_operator_library_binds_only_the_public_abi(sq):
    """libsvdq_torch.so (TORCH_LIBRARY registrations of torch.ops.svdq.*) reaches the kernels through include/svdq.h
    alone: every svdq_* symbol it imports is a declared entry point, and it defines none of its own."""
    path = sq.torch_ops.OPS_LIB_PATH
    assert os.path.exists(path)
    und = subprocess.check_output(["nm", "-D", "--undefined-only", path], text=True)
    used = sorted({ln.split()[-1] for ln in und.splitlines() if " U svdq_" in ln})
    assert used and set(used) <= set(_header_functions()), used
    for must in ("svdq_compress", "svdq_rtvq_quantize", "svdq_mask_combine", "svdq_ingest", "svdq_task_gram"):
        assert must in used
    defd = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    assert not [ln for ln in defd.splitlines() if " T svdq_" in ln]
    needed = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-d", path], text=True)
    assert "libsvdq_hip.so" in needed and "$ORIGIN" in needed


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "svdq.h")).read()
    for ref_file in ("basis.py", "compress.py", "rtvq.py", "mask_loader.py", "cli.py", "config.py",
                     "quantization_utils.py"):
        assert ref_file in text, ref_file


def test_library_targets_gfx950(sq):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", sq._native.LIB_PATH],
                         capture_output=True, text=True).stdout
    blob = open(sq._native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob[:0]  # single-target build


def test_python_surface_mirrors_reference(sq):
    """names, argument orders and defaults of SURVEY.md section 8(b)."""
    def params(fn):
        return [(p.name, p.default) for p in inspect.signature(fn).parameters.values()]
    E = inspect.Parameter.empty
    assert params(sq.construct_basis) == [("deltas", E), ("energy_threshold", 0.90), ("max_rank", None),
                                          ("center", True), ("device", "cpu"), ("use_randomized", False),
                                          ("verbose", True)]
    assert params(sq.construct_masked_basis) == [("masked_deltas", E), ("unmasked_deltas", E),
                                                 ("energy_threshold", 0.90), ("max_rank", None), ("center", True),
                                                 ("device", "cpu"), ("include_noise", False), ("verbose", False)]
    assert params(sq.select_rank) == [("singular_values", E), ("energy_threshold", 0.90), ("max_rank", None),
                                      ("min_rank", 1)]
    assert params(sq.compress_single_task) == [("task_delta", E), ("U_high", E), ("U_low", E), ("quantizer", E),
                                               ("device", "cpu"), ("mean", None)]
    assert params(sq.compress_all_parameters) == [("task_vectors", E), ("masks", E), ("bases", E), ("config", E),
                                                  ("device", "cpu")]
    assert params(sq.RTVQQuantizer.__init__)[1:] == [("num_bits", 4), ("num_stages", 2)]
    assert params(sq.asymmetric_quantization) == [("X", E), ("qbit", 8), ("verbose", False)]
    assert params(sq.multistage_residual_quantization) == [("tensor", E), ("num_bits", 4), ("num_stages", 2),
                                                           ("verbose", False)]
    assert params(sq.multistage_residual_dequantization) == [("payloads", E), ("device", "cpu")]
    assert params(sq.combine_masks) == [("task_masks", E), ("strategy", "union"), ("device", "cpu"),
                                        ("verbose", True)]
    assert params(sq.reconstruct_from_coefficients) == [("avg_c_high", E), ("avg_c_low", E), ("U_high", E),
                                                        ("U_low", E), ("device", "cpu"), ("mean", None)]


def test_config_defaults_and_validation(sq):
    c = sq.SVDHybridConfig()
    assert (c.svd_energy_threshold, c.svd_max_rank, c.svd_center, c.svd_fp16) == (0.95, 64, True, True)
    assert (c.svd_low_bits, c.svd_rtvq_stages, c.svd_mask_strategy, c.svd_include_noise) == (4, 2, "union", False)
    assert (c.svd_noise_shrink, c.svd_min_mask_size, c.svd_weighting, c.device) == (0.5, 10, "uniform", "cuda")
    for bad in (dict(svd_mask_strategy="xor"), dict(svd_weighting="best"), dict(svd_energy_threshold=0.0),
                dict(svd_energy_threshold=1.5), dict(svd_low_bits=0), dict(svd_low_bits=9), dict(svd_rtvq_stages=0)):
        with pytest.raises(ValueError):
            sq.SVDHybridConfig(**bad)


def test_host_side_rank_rule_kats(sq):
    """select_rank / compute_energy_spectrum are host arithmetic on N scalars: the reference's KATs
    (tests/test_rank_selection.py:27-77) hold without a GPU."""
    assert sq.select_rank(torch.tensor([10.0, 1e-10, 1e-12]), 0.99) == 1
    assert sq.select_rank(torch.ones(100), 0.99, max_rank=10) <= 10
    assert sq.select_rank(torch.tensor([100.0, 0.01, 0.001]), 0.999, None, min_rank=2) >= 2
    cum = sq.compute_energy_spectrum(torch.tensor([4.0, 3.0, 2.0, 1.0]))
    assert abs(cum[-1].item() - 1.0) < 1e-6 and all(cum[i] <= cum[i + 1] for i in range(3))
    S = torch.tensor([10.0, 5.0, 2.0, 1.0, 0.5])
    assert sq.select_rank(S, 0.5) <= sq.select_rank(S, 0.9) <= sq.select_rank(S, 0.99)
    with pytest.raises(ValueError):
        sq.stack_and_center([], True)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_not_silently(sq):
    with pytest.raises(RuntimeError):
        sq.RTVQQuantizer(4, 2).quantize(torch.randn(16))
    with pytest.raises(RuntimeError):
        sq.construct_basis([torch.randn(64) for _ in range(4)], verbose=False)
    with pytest.raises(RuntimeError):
        sq.compute_union_mask([torch.ones(4, dtype=torch.bool)])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "svd-quantization-task-merging_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, f)


def test_main_dispatcher_unimplemented_methods():
    """reference src/main.py:73-89: the three listed-but-unimplemented methods print a notice and return None."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for method in ("task_arithmetic", "ties", "dare"):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "main.py"), "--method", method],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and f"Method '{method}' not yet implemented" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "main.py"), "--method", "nope"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_every_public_callable_of_the_reference_modules_is_offered(sq):
    """tests/golden/api_signatures.json was written by make_golden.py from the imported reference: every public
    function / class / method of its in-scope modules, with argument names and defaults.  This package must offer each
    one under the same name with the same leading arguments and defaults (it may add trailing ones)."""
    import json
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")))
    home = {"reload": "storage"}                       # reload.py's two functions live in storage.py here

    def sig(fn, drop_self=False):
        out = []
        for name, prm in list(inspect.signature(fn).parameters.items())[1 if drop_self else 0:]:
            if prm.kind in (prm.VAR_POSITIONAL, prm.VAR_KEYWORD):
                out.append(("*" if prm.kind == prm.VAR_POSITIONAL else "**") + name)
            else:
                out.append(name if prm.default is prm.empty else f"{name}={prm.default!r}")
        return out

    import importlib
    problems = []
    for mod_name, entries in api.items():
        mod = importlib.import_module(f"svdq_amd.{home.get(mod_name, mod_name)}")
        for name, want in entries.items():
            cls, _, meth = name.partition(".")
            obj = getattr(mod, cls, None)
            if obj is None:
                problems.append(f"{mod_name}.{name}: missing")
                continue
            if meth:
                obj = getattr(obj, meth, None)
                if obj is None:
                    problems.append(f"{mod_name}.{name}: missing")
                    continue
                got = sig(obj, drop_self=True)
            elif inspect.isclass(obj):
                got = sig(obj.__init__, drop_self=True)
            else:
                got = sig(obj)
            # an optional argument where the reference has a required one is a superset (reload's output_path)
            same = len(got) >= len(want) and all(g == w or g.split("=")[0] == w for g, w in zip(got, want))
            if not same:
                problems.append(f"{mod_name}.{name}: reference {want}, here {got}")
    assert not problems, "\n".join(problems)


def test_hydra_style_configuration(sq):
    """hydra_entry.py:66-101: nested configuration -> SVDHybridConfig with the entry point's own defaults."""
    from svdq_amd import hydra_entry
    cfg = hydra_entry.config_from_hydra({"tasks": ["Cars", "DTD"], "checkpoint_dir": "ck", "method": {"svd_low_bits": 2}})
    assert cfg.tasks == ["Cars", "DTD"] and cfg.checkpoint_dir == "ck" and cfg.svd_low_bits == 2
    assert cfg.svd_energy_threshold == 0.90 and cfg.svd_max_rank == 128 and cfg.svd_weighting_temperature == 1.0
    assert cfg.output_dir == "./svd_hybrid_output" and cfg.device == "cuda"
    assert hydra_entry.config_from_hydra({"method": "svd_hybrid"}).svd_rtvq_stages == 2
    with pytest.raises(ValueError):
        hydra_entry.config_from_hydra({"method": {"svd_mask_strategy": "nope"}})
