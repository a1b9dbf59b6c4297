"""torch.ops.svdq.* -- the custom-operator face of the C ABI (BASELINE north_star)."""
import numpy as np
import pytest
import torch

import svdq_amd  # noqa: F401  (registers the operators when the libraries are already built)

OPS = ("rtvq_quantize", "rtvq_dequantize", "mask_combine", "mask_select", "compress", "ingest", "task_gram",
       "compress_masked", "compress_gather", "compress_from_base", "mask_combine_indices", "reconstruct", "recon_error",
       "merge", "merge_masked", "diagnostics")


@pytest.fixture(scope="module", autouse=True)
def _ops_loaded():
    """A fresh checkout has no libraries when pytest imports this module: build them, then register the operators."""
    import os
    from svdq_amd import torch_ops, _native
    if not (os.path.exists(_native.LIB_PATH) and os.path.exists(torch_ops.OPS_LIB_PATH)):
        _native.build()
    torch_ops.load()


def test_ops_are_registered_natively():
    """The operators come from libsvdq_torch.so (TORCH_LIBRARY in csrc/svdq_torch.cpp), which is linked against
    the in-tree libsvdq_hip.so; no Python function stands behind any of them."""
    import os
    from svdq_amd import torch_ops, _native
    assert os.path.exists(torch_ops.OPS_LIB_PATH)
    mapped = open("/proc/self/maps").read()
    assert os.path.realpath(torch_ops.OPS_LIB_PATH) in mapped and os.path.realpath(_native.LIB_PATH) in mapped
    assert mapped.count("libsvdq_hip.so") and len({l.split()[-1] for l in mapped.splitlines()
                                                   if l.endswith("libsvdq_hip.so")}) == 1   # one copy, not two
    for name in OPS:
        schema = getattr(torch.ops.svdq, name).default._schema
        assert schema.name == f"svdq::{name}"
        # a Python-registered kernel shows up in torch.library's registry of the process; none may
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"svdq::{name}", "CPU")
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"svdq::{name}", "CUDA")
    assert torch_ops.plan_cache_size() >= 0


def test_ops_are_registered_and_have_no_cpu_kernel():
    for name in OPS:
        assert hasattr(torch.ops.svdq, name), name
    # there is no CPU implementation to fall back to: CPU tensors are refused by the dispatcher
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.svdq.rtvq_quantize(torch.randn(16), 4, 2)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.svdq.mask_combine([torch.ones(4, dtype=torch.bool)], "union")


@pytest.mark.gpu
def test_ops_match_the_python_layer():
    from oracle import svd_hybrid_oracle as orc
    from svdq_amd.pipeline import CompressPlan
    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    x = 0.02 * torch.randn(100003, device=dev)
    codes, scale, zp, rn = torch.ops.svdq.rtvq_quantize(x, 4, 2)
    want = orc.rtvq_quantize(x.cpu().numpy(), 4, 2)
    assert np.array_equal(codes.cpu().numpy(), want["codes"])
    assert np.array_equal(scale.cpu().numpy(), want["scale"]) and np.array_equal(zp.cpu().numpy(), want["zero_point"])
    deq = torch.ops.svdq.rtvq_dequantize(codes, scale, zp)
    assert np.array_equal(deq.cpu().numpy(), orc.rtvq_dequantize(want).reshape(-1))
    masks = [torch.rand(5000, device=dev) > 0.5 for _ in range(3)]
    u = torch.ops.svdq.mask_combine(masks, "union")
    assert torch.equal(u, masks[0] | masks[1] | masks[2])
    assert torch.equal(torch.ops.svdq.mask_select(x[:5000], u, False), x[:5000][u])
    assert torch.equal(torch.ops.svdq.mask_select(x[:5000], u, True), x[:5000][~u])
    N, sizes = 8, [70001, 768]
    vecs = [[d.to(dev) for d in orc.synthetic_deltas(D, N, 40 + i)] for i, D in enumerate(sizes)]
    small, basis, mean = torch.ops.svdq.compress([v for vs in vecs for v in vs], N, 0.9, 0, True, True, 4, 2)
    plan = CompressPlan(sizes, N, energy_threshold=0.9, max_rank=None, center=True, fp16=True, low_bits=4,
                        rtvq_stages=2, device=dev)
    plan.run(plan.pointer_table(vecs))
    torch.cuda.synchronize()
    assert torch.equal(small, plan.small) and mean.numel() == plan.mean.numel()
    # a second call with the same shapes reuses the cached plan: no new plan, fresh outputs, same bits, and the
    # first call's outputs are untouched
    from svdq_amd import torch_ops
    n_plans = torch_ops.plan_cache_size()
    small_copy = small.clone()
    small2, basis2, mean2 = torch.ops.svdq.compress([v for vs in vecs for v in vs], N, 0.9, 0, True, True, 4, 2)
    assert torch_ops.plan_cache_size() == n_plans and small2.data_ptr() != small.data_ptr()
    torch.cuda.synchronize()
    assert torch.equal(small2, small_copy) and torch.equal(small, small_copy)
    for p, D in enumerate(sizes):              # the packed buffers have uninitialised alignment gaps: compare views
        o = plan.mean_off[p]
        assert torch.equal(mean2[o:o + D], mean[o:o + D])
    sm = plan.fetch_small()
    for p, D in enumerate(sizes):
        a = plan.basis_tensors(p, int(sm.k[p]), int(sm.r[p]), D)
        lo = plan.slab_off[p]
        assert torch.equal(basis[lo:lo + a[0].numel() * 2].view(torch.float16).view(a[0].shape), a[0])
    base = torch.randn(33, 7, device=dev)
    fts = [base + 0.1 * torch.randn_like(base) for _ in range(3)]
    for d, f in zip(torch.ops.svdq.ingest(base, fts), fts):
        assert torch.equal(d, f - base)
    G = torch.ops.svdq.task_gram([v for vs in vecs for v in vs], N)
    ref = sum(torch.stack(vs).double() @ torch.stack(vs).double().T for vs in vecs)
    assert torch.allclose(G, ref, rtol=2e-6, atol=2e-6 * float(ref.abs().max()))


@pytest.mark.gpu
def test_masked_from_base_and_consumer_ops_match_the_ctypes_route():
    """compress_masked / compress_gather / compress_from_base / mask_combine_indices / reconstruct / recon_error / merge /
    merge_masked / diagnostics: every operator against the same entry point reached through ctypes (svdq_amd.pipeline),
    bit for bit."""
    from oracle import svd_hybrid_oracle as orc
    from svdq_amd.pipeline import CompressPlan
    from svdq_amd.mask_loader import MaskSet
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(8)
    N, sizes = 6, [70001, 777, 4096 * 5]
    vecs = [[d.to(dev) for d in orc.synthetic_deltas(D, N, 60 + i)] for i, D in enumerate(sizes)]
    flat = [v for vs in vecs for v in vs]
    masks = [(torch.rand(D, generator=g) < 0.85).to(dev) for D in sizes]
    kw = dict(energy_threshold=0.9, max_rank=None, center=True, fp16=True, low_bits=4, rtvq_stages=2, device=dev)
    ms = MaskSet(sizes, dev)
    ct, _ = ms.count_scan(masks)
    mtab = torch.tensor([m.data_ptr() for m in ms._s["mb"]], dtype=torch.int64).to(dev)
    ref = CompressPlan(sizes, N, **kw)
    ref.run_masked(ref.pointer_table(vecs), mtab, ms.unit_starts(ref, ct), ct)
    torch.cuda.synchronize()

    def same_artifacts(small, basis, mean, plan):
        assert torch.equal(small, plan.small)
        sm = plan.fetch_small()
        for p in range(len(sizes)):
            rows, k, r = int(sm.rows[p]), int(sm.k[p]), int(sm.r[p])
            a = plan.basis_tensors(p, k, r, rows)
            lo = plan.slab_off[p]
            assert torch.equal(basis[lo:lo + a[0].numel() * 2].view(torch.float16).view(a[0].shape), a[0])
            o = plan.mean_off[p]
            assert torch.equal(mean[o:o + rows], a[2].flatten())
    for op in (torch.ops.svdq.compress_masked, torch.ops.svdq.compress_gather):
        small, basis, mean, rows = op(flat, masks, N, 0.9, 0, True, True, 4, 2)
        torch.cuda.synchronize()
        assert torch.equal(rows, ct)
        same_artifacts(small, basis, mean, ref)
    with pytest.raises(ValueError, match="Shape mismatch"):
        torch.ops.svdq.compress_masked(flat, [masks[0], masks[1], masks[2][:5]], N, 0.9, 0, True, True, 4, 2)
    # straight from checkpoints
    base = [torch.randn(D, generator=g).to(dev) for D in sizes]
    ft = [[base[p] + vecs[p][t] for t in range(N)] for p in range(len(sizes))]
    fb = CompressPlan(sizes, N, **kw)
    fb.run_from_base(fb.pointer_table(ft), torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev))
    small, basis, mean = torch.ops.svdq.compress_from_base([f for fs in ft for f in fs], base, N, 0.9, 0, True, True, 4, 2)
    torch.cuda.synchronize()
    same_artifacts(small, basis, mean, fb)
    # combine + index lists
    per_task = [[(torch.rand(D, generator=g) > 0.6).to(dev) for _ in range(3)] for D in sizes]
    comb, idx, cnt = torch.ops.svdq.mask_combine_indices([m for ms_ in per_task for m in ms_], 3, "majority")
    for q, D in enumerate(sizes):
        want = torch.stack([m.int() for m in per_task[q]]).sum(0) * 2 >= 3
        assert torch.equal(comb[q], want) and int(cnt[q]) == int(want.sum())
        assert torch.equal(idx[q][:int(cnt[q])].long(), torch.nonzero(want).flatten())
    # consumers of one basis
    sm = fb.fetch_small()
    k, r = int(sm.k[0]), int(sm.r[0])
    Uh, Ul, mu = fb.basis_tensors(0, k, r, sizes[0])
    coef = torch.randn(r, generator=g).to(dev)
    import svdq_amd
    want = svdq_amd.merge._reconstruct(coef[:k], coef[k:], Uh, Ul, mu, 0.5)
    assert torch.equal(torch.ops.svdq.reconstruct(Uh, Ul, coef, mu, 0.5), want)
    assert torch.equal(torch.ops.svdq.reconstruct(Uh, Ul, coef, None, 1.0), svdq_amd.merge._reconstruct(coef[:k], coef[k:], Uh, Ul, None, 1.0))
    x = ft[0][2] - base[0]
    e6 = torch.ops.svdq.recon_error(Uh, Ul, coef, None, x)
    ref6 = svdq_amd.diagnostics._fused_error(x, Uh, Ul, coef[:k], coef[k:], dev)
    assert [float(v) for v in e6.cpu()] == [ref6[key] for key in svdq_amd.diagnostics._KEYS]
    from helpers import diag_check      # and the operator's numbers against diagnostics.py:186-215 in fp64 on the same operands
    diag_check(ref6, x.cpu(), Uh.cpu(), Ul.cpu(), coef[:k].cpu(), coef[k:].cpu(), what="torch.ops.svdq.recon_error")
    # the plan-level merge on the buffers the compress operator returned
    w = torch.tensor([0.3, 0.1, 0.2, 0.15, 0.05, 0.2], device=dev)
    outs = torch.ops.svdq.merge(small, basis, mean, sizes, N, 0.9, 0, True, True, 4, 2, w, base)
    buf, offs = fb.merge(w.view(1, N), base_table=torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev))
    torch.cuda.synchronize()
    for p, D in enumerate(sizes):
        assert torch.equal(outs[p], buf[offs[p]:offs[p] + D])
    # a mean buffer that is not the plan's (short, or another dtype) is refused before any kernel reads past its end
    with pytest.raises((ValueError, RuntimeError), match="mean must be the plan's float32 mean buffer"):
        torch.ops.svdq.merge(small, basis, mean[:10].contiguous(), sizes, N, 0.9, 0, True, True, 4, 2, w, base)
    with pytest.raises((ValueError, RuntimeError), match="mean must be the plan's float32 mean buffer"):
        torch.ops.svdq.merge(small, basis, mean.double(), sizes, N, 0.9, 0, True, True, 4, 2, w, base)
    # plan-level diagnostics of the same buffers (the deltas of the from-base run are fine-tuned minus base)
    d6 = torch.ops.svdq.diagnostics([f - base[p] for p, fs in enumerate(ft) for f in fs], [], small, basis, mean, N,
                                    0.9, 0, True, True, 4, 2, False)
    keepd = [[f - base[p] for f in fs] for p, fs in enumerate(ft)]
    want6 = fb.diagnostics(fb.pointer_table(keepd))
    torch.cuda.synchronize()
    assert torch.equal(d6, want6)
    # masked: the buffers of compress_masked, merged back to full size / diagnosed through the mask
    small_m, basis_m, mean_m, rows_m = torch.ops.svdq.compress_masked(flat, masks, N, 0.9, 0, True, True, 4, 2)
    outs_m = torch.ops.svdq.merge_masked(small_m, basis_m, mean_m, masks, N, 0.9, 0, True, True, 4, 2, w, base)
    cbuf, coffs = ref.merge(w.view(1, N), rows_dev=ct)
    torch.cuda.synchronize()
    from svdq_amd.mask_loader import reconstruct_from_masked
    for p, D in enumerate(sizes):
        full = reconstruct_from_masked(cbuf[coffs[p]:coffs[p] + int(ct[p])], None, masks[p], masks[p].shape)
        assert torch.equal(outs_m[p], base[p] + full)
    dm = torch.ops.svdq.diagnostics(flat, masks, small_m, basis_m, mean_m, N, 0.9, 0, True, True, 4, 2, False)
    wantm = ref.diagnostics_masked(ref.pointer_table(vecs), mtab, ms.unit_starts(ref, ct), ct)
    torch.cuda.synchronize()
    assert torch.equal(dm, wantm)
    # a mask on another device than the first is refused before any kernel runs (one GPU here: the CPU stands in)
    with pytest.raises((ValueError, RuntimeError, NotImplementedError)):
        torch.ops.svdq.mask_combine([masks[0], masks[0].cpu()], "union")
