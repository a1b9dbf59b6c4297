"""Randomised parity cases shared by the GPU test-suite (a seeded subset) and tools/fuzz_*.py (long runs).

oracle_case : random (D, N, thresholds, centre, fp16, bits, stages) against the CPU oracle (the restated reference op
              sequence): singular values, rank, retained energy, reconstructions; and, free of the basis freedom,
              the span of U_high (principal angles against the reference's, where a gap makes it unique), the
              projections on the device's own basis (fp64 on the host) and the reference quantizer on the device's
              own c_low, bit for bit.
modes_case  : gather mode (against compacted copies), minus-base mode (against ingest + compress), both combined
              (against ingest + gather) and the mask walk must reproduce the plain path bit for bit; the masked
              consumers (svdq_merge_masked, svdq_diagnostics_masked) must reproduce merge / diagnostics of the
              compacted copies.
diag_case   : the diagnostics kernels (svdq_diagnostics, svdq_diagnostics_masked in both polarities, svdq_recon_error)
              against diagnostics.py:186-215 evaluated in fp64 on the artifacts the kernel read -- random task counts,
              ragged sizes, fp16 / fp32, centred or not, unit sizes, mask densities, spikes at random rows.
merge_case  : the plan-level merge (svdq_merge: coefficient averaging + streaming reconstruction, sets / shares / scale /
              + base) against merge.py:61-194, 429-552 evaluated in fp64 on the plan's own artifacts, element by element.
Each returns a (description, [mismatch messages]) pair; an empty list means the case is within tolerance."""
import random

import numpy as np
import torch

SIGMA_FLOOR = 1e-5   # sigma / sigma_0 above which singular values are compared (LAPACK's own fp32 error reaches 3e-5
SIGMA_RTOL = 1e-4    # relative at 1e-5 sigma_0; exact-product Gram sums are closer to the fp64 values than that)


SIGN_ENUM = 8        # up to this many quantized directions the sign choices are enumerated (2^8 quantizer calls per task)


def _worst_sign_error(orc, ref, er2: float, bits: int, stages: int) -> float:
    """Largest squared reconstruction error (summed over the tasks) the REFERENCE's own pipeline reaches over the sign
    choices of its quantized basis columns; everything but the quantizer's input signs is held fixed."""
    import itertools
    lows = [t["c_low"].numpy().astype(np.float32) for t in ref["tasks"]]
    nl = lows[0].size

    def qerr(signs):
        tot = 0.0
        for c in lows:
            x = c * signs
            tot += float(np.sum((x - orc.rtvq_dequantize(orc.rtvq_quantize(x, bits, stages)).reshape(-1)) ** 2))
        return tot
    own = qerr(np.ones(nl, np.float32))
    rest = max(er2 - own, 0.0)                   # fp16 rounding of the basis and of c_high: the same for every choice
    # a global flip mirrors the quantizer's grid with the data, so the first sign is held at +1
    worst = max(qerr(np.array((1.0,) + s, np.float32)) for s in itertools.product((1.0, -1.0), repeat=nl - 1))
    return rest + worst


def oracle_case(sq, orc, dev, seed: int, c: int):
    rnd = random.Random(1000003 * seed + c)
    N = rnd.choice([1, 2, 3, 4, 6, 8, 8, 11, 16, 17, 20, 27, 32])
    D = rnd.choice([1, 2, 5, 31, 255, 256, 257, 1000, 4099, rnd.randint(1, 60000), rnd.randint(1, 200000)])
    thr = rnd.choice([0.5, 0.9, 0.95, 0.999, 1.0])
    max_rank = rnd.choice([None, 1, 3, 64])
    center, fp16 = rnd.random() < 0.75, rnd.random() < 0.75
    bits, stages = rnd.choice([2, 4, 8]), rnd.choice([1, 2, 3])
    desc = f"D={D} N={N} thr={thr} max_rank={max_rank} center={center} fp16={fp16} b={bits} S={stages}"
    deltas = orc.synthetic_deltas(D, N, 5000 + 31 * seed + c, rank=min(3, N))
    ref = orc.compress_parameter(deltas, thr, max_rank, center, fp16, bits, stages)
    plan, sm = sq.compress_batch([[d.to(dev) for d in deltas]], energy_threshold=thr, max_rank=max_rank, center=center,
                                 fp16=fp16, low_bits=bits, rtvq_stages=stages, device=dev)
    k, r = int(sm.k[0]), int(sm.r[0])
    msgs = []
    S_ref = ref["basis"]["singular_values"].numpy()
    if r != len(S_ref):
        return desc, [f"r {r} vs {len(S_ref)}"]
    real = S_ref > SIGMA_FLOOR * max(S_ref[0], 1e-30)
    if not np.allclose(sm.sigma[0, :r][real], S_ref[real], rtol=SIGMA_RTOL):
        msgs.append(f"sigma {sm.sigma[0, :r]} vs {S_ref}")
    e = S_ref.astype(np.float32) ** 2
    cum = np.cumsum(e, dtype=np.float32) / max(e.sum(dtype=np.float32), 1e-30)
    near = np.any(np.abs(cum - thr) < 1e-4)
    if not near and k != ref["basis"]["k"]:
        msgs.append(f"k {k} vs {ref['basis']['k']}")
    # Free of the basis freedom (signs, rotations inside near-degenerate subspaces), so checked to tight tolerances in every
    # case: the coefficients are the projections on the DEVICE's own (rounded) basis, and the device's payloads are
    # what the reference quantizer (the C oracle) makes of the device's own c_low, bit for bit.
    if r > 0:
        Uh, Ul, mu = plan.basis_tensors(0, k, r, D)
        U = torch.cat([Uh, Ul], dim=1).double().cpu().numpy() if (k > 0 and r > k) else \
            (Uh if k > 0 else Ul).double().cpu().numpy()
        mu32 = mu.flatten().cpu().numpy() if mu is not None else None
        for t in range(N):
            x32 = deltas[t].numpy() - mu32 if mu32 is not None else deltas[t].numpy()      # fp32, like compress.py:44
            want = U.T @ x32.astype(np.float64)
            got = sm.coef[0, t, :r].astype(np.float64)
            # column i is U_i = Tc v_i / sigma_i formed in fp32: its own error is ~eps32 sigma_0 / sigma_i, and the
            # coefficient (closed form sigma_i v_i + the fp16-rounding correction) differs from the explicit projection
            # on the stored column by that much of |x| (seed 5001 case 805: sigma_30 = 5e-4 sigma_0, 3e-5 |x|)
            # The model has no ceiling of its own: round 3 had capped it at 1e-3 |x| by hand, which a direction at
            # 1.9e-6 sigma_0 -- above the 1e-6 null threshold, below the 1e-5 floor under which not even singular values
            # are compared; its fp32-formed column is only ~97 % the singular vector -- exceeded with 1.75e-3 |x| (seed 4
            # case 37: D = 31 < N = 32; model: 0.16 |x|).  Capped where the model reaches 1e-2 (sigma_i = 3e-5 sigma_0);
            # such a direction holds < 1e-9 of the energy and the reconstructions still agree to MSE 3e-10.
            sg = np.maximum(sm.sigma[0, :r].astype(np.float64), 1e-30)
            tol = float(np.linalg.norm(x32)) * np.minimum(1e-5 + 3e-7 * sg[0] / sg, 1e-2) + 1e-12
            if mu32 is not None:      # the rows of T - mean sum to a few ulp(mean), not to 0; the device removes that
                tol = tol + 2.4e-7 * float(np.linalg.norm(mu32))      # component (1 / sqrt(N) is deflated), a GEMV keeps it
            if np.isfinite(want).all() and not np.all(np.abs(got - want) <= tol):
                bad_i = int(np.argmax(np.abs(got - want) / tol))
                msgs.append(f"projection on the device's basis: task {t} column {bad_i}, |dc| "
                            f"{abs(got[bad_i] - want[bad_i]):.2e} (tol {tol[bad_i]:.1e})")
                break
            if r > k:
                art = sq.pipeline.task_artifact(plan, sm, 0, t)["c_low_quant"]
                oq = orc.rtvq_quantize(sm.coef[0, t, k:r].copy(), bits, stages)
                same = len(art["payloads"]) == oq["codes"].shape[0]
                for s_, pa in enumerate(art["payloads"]):
                    if not same:
                        break
                    qa = pa["quantized"].numpy().reshape(-1)
                    sa = np.array([pa["scale"].item(), pa["zero_point"].item()], dtype=np.float32)
                    sb = np.array([oq["scale"][s_], oq["zero_point"][s_]], dtype=np.float32)
                    # (a NaN is a NaN: x86 and gfx950 differ in the sign bit of the default NaN -- the F4 cases)
                    same = np.array_equal(qa, oq["codes"][s_]) and bool(np.all(
                        (sa.view(np.uint32) == sb.view(np.uint32)) | (np.isnan(sa) & np.isnan(sb))))
                if not same:
                    msgs.append(f"quantizer on the device's own c_low: task {t} differs from the reference quantizer")
                    break
    # the span of the high-energy columns is unique when a gap separates sigma_{k-1} from sigma_k: principal angles between
    # the reference's U_high and the device's (both fp16-rounded: cosines within ~1e-3 of 1)
    if k == ref["basis"]["k"] and 0 < k < r and real[k - 1] and S_ref[k - 1] > 1.05 * S_ref[k] and D > k:
        Ud = plan.basis_tensors(0, k, r, D)[0].float().cpu().numpy().astype(np.float64)
        Ur = ref["basis"]["U_high"].float().numpy().astype(np.float64)
        if np.isfinite(Ud).all() and np.isfinite(Ur).all():
            gap = float(S_ref[k - 1] / S_ref[k]) - 1.0
            cosines = np.linalg.svd(Ur.T @ Ud, compute_uv=False)
            # fp32 SVDs leave ~eps32 / gap of mixing between the two sides of the gap, fp16 rounding 1e-3 of norm
            if cosines.min() < 1.0 - 2e-3 - min(1e-5 / gap, 0.5):
                msgs.append(f"span of U_high: smallest principal cosine {cosines.min():.6f} (gap {gap:.2e})")
    if k == ref["basis"]["k"] and r - k > 2:
        U_high, U_low, mean = plan.basis_tensors(0, k, r, D)
        quant = sq.RTVQQuantizer(bits, stages)
        eo2 = er2 = 0.0
        coarse = bits <= 2
        for t in range(N):
            art = sq.pipeline.task_artifact(plan, sm, 0, t)
            cl = quant.dequantize(art["c_low_quant"], device=dev).float()
            rec = sq.reconstruct_from_coefficients(art["c_high_fp16"].to(dev).float(), cl, U_high, U_low, dev,
                                                   mean=mean).cpu().numpy()
            rr = ref["recon"][t].numpy()
            if np.isfinite(rr).all() and np.isfinite(rec).all():
                x = deltas[t].numpy()
                mse = float(np.mean((rec - rr) ** 2))
                ref_noise = float(np.mean((rr - x) ** 2))      # the reference's own error against the original
                eo2 += float(np.linalg.norm(rec - x) ** 2)
                er2 += float(np.linalg.norm(rr - x) ** 2)
                if mse > 1e-6:
                    # Contract for quantization-noise-dominated cases: the basis inside near-degenerate singular
                    # subspaces (and every sign) is not unique and the min/max quantizer is not invariant under that
                    # freedom, so where the reference's OWN reconstruction is 1e-7 or more (MSE) off the original --
                    # 2 bits always; 4 bits x 1 stage when max_rank pushes a large direction into c_low -- two
                    # equally valid bases differ by that noise.  Then the error against the ORIGINAL deltas is compared,
                    # aggregated over the tasks.  Everywhere else the two reconstructions agree to MSE <= 1e-6.
                    if bits > 2 and ref_noise <= 1e-7:
                        msgs.append(f"recon mse {mse:.2e} task {t} (reference noise {ref_noise:.1e})")
                        break
                    coarse = True
        # How far two valid bases can differ in this criterion depends on the freedom they have: inside a cluster of
        # near-equal singular values among the quantized directions ANY rotation is a valid basis (the structured
        # generator's noise tail is such a cluster) -- observed 0.74x .. 1.92x over 60 configurations, bound 2.5x; with
        # every quantized direction separated by a clear gap only the signs are free: bound 1.5x.
        # With at most SIGN_ENUM quantized directions the sign freedom is enumerated instead of bounded: the basis is
        # orthonormal, so a sign choice s changes the error by sum_t |c_low s - dequantize(quantize(c_low s))|^2 only
        # (fuzz seed 304 case 144: three directions, 2 bits -> 2.65e-4 .. 4.67e-4 rms over the eight choices, and
        # the device's basis lands on one of them; seed 5001 case 807: five directions with a near-degenerate pair,
        # 5.0e-4 .. 1.78e-3 over the sign choices alone -- the reference's own basis being the best of them -- and up
        # to 3.2e-3 with rotations inside the pair).  The worst choice, with 10 % on top for the basis rounding, is the
        # bound there.
        low = S_ref[k:r][real[k:r]]
        clustered = low.size >= 2 and bool(np.any(np.abs(np.diff(low)) < 0.05 * low[:-1]))
        bound = 2.5 if clustered else 1.5
        limit = bound ** 2 * er2
        if coarse and r - k <= SIGN_ENUM:      # (a cluster only adds rotations on top of the sign choices)
            limit = max(limit, 1.1 ** 2 * _worst_sign_error(orc, ref, er2, bits, stages))
        if coarse and eo2 > limit + 1e-12:
            msgs.append(f"rms recon error vs original {eo2 ** 0.5:.3e} vs reference {er2 ** 0.5:.3e} "
                        f"(limit {limit ** 0.5:.3e})")
    plan.close()
    return desc, msgs


def _same(a, b):
    if not torch.equal(a.small, b.small):
        return "small buffers differ"
    sm = a.fetch_small()
    for p in range(a.P):
        rows = int(sm.rows[p])
        x = a.basis_tensors(p, int(sm.k[p]), int(sm.r[p]), rows)
        y = b.basis_tensors(p, int(sm.k[p]), int(sm.r[p]), rows)
        for u, v in zip(x, y):
            if (u is None) != (v is None) or (u is not None and not torch.equal(u, v)):
                return f"basis/mean of parameter {p} differs"
    return None


def modes_case(sq, dev, seed: int, c: int):
    from svdq_amd.pipeline import CompressPlan
    from svdq_amd.mask_loader import MaskSet
    rnd = random.Random(7000003 * seed + c)
    g = torch.Generator(device=dev).manual_seed(97 * seed + c)
    N = rnd.choice([1, 2, 3, 5, 8, 8, 8, 12, 16, 17, 20, 24, 32])
    P = rnd.randint(1, 5)
    sizes = [rnd.choice([1, 3, 7, 255, 256, 257, 1000, 4096, 5001, 65536 + rnd.randint(0, 9), rnd.randint(1, 300000)])
             for _ in range(P)]
    fp16, center = rnd.random() < 0.7, rnd.random() < 0.8
    bits, stages = rnd.choice([2, 4, 8]), rnd.choice([1, 2, 4])
    unit_rows = rnd.choice([0, 1024, 4096])
    dens = rnd.choice([0.0, 0.05, 0.5, 0.94, 1.0])
    desc = f"N={N} sizes={sizes} fp16={fp16} center={center} bits={bits} stages={stages} dens={dens}"
    kw = dict(energy_threshold=rnd.choice([0.5, 0.9, 0.99]), max_rank=rnd.choice([None, 2, 64]), center=center, fp16=fp16,
              low_bits=bits, rtvq_stages=stages, device=dev, unit_rows=unit_rows)
    base = [torch.randn(D, device=dev, generator=g) for D in sizes]
    lat = [torch.randn(D, 3, device=dev, generator=g) for D in sizes]
    deltas = [[0.01 * (lat[p] @ torch.randn(3, device=dev, generator=g))
               + 0.002 * torch.randn(sizes[p], device=dev, generator=g) for _ in range(N)] for p in range(P)]
    msgs = []
    # minus-base: fine-tuned = base + delta is not exactly invertible in fp32, so compare with ingest of the same tensors
    fts = [[base[p] + deltas[p][t] for t in range(N)] for p in range(P)]
    eb = sq.ElementwiseBatch(sizes, N, dev)
    ing = eb.ingest(base, [f for fs in fts for f in fs])
    r2 = CompressPlan(sizes, N, **kw)
    r2.run(r2.pointer_table([ing[p * N:(p + 1) * N] for p in range(P)]))
    fb = CompressPlan(sizes, N, **kw)
    fb.run_from_base(fb.pointer_table(fts), torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev))
    torch.cuda.synchronize()
    m = _same(r2, fb)
    if m:
        msgs.append("from_base: " + m)
    eb.close()
    # gather vs compaction
    masks = [(torch.rand(D, device=dev, generator=g) < dens) for D in sizes]
    ms = MaskSet(sizes, dev)
    dt, _, ct, _ = ms.compact(masks, deltas, want_false=False)
    it, _, ct2, _ = ms.indices(masks, want_false=False)
    r3 = CompressPlan(sizes, N, **kw)
    r3.run(r3.pointer_table(dt), ct)
    ga = CompressPlan(sizes, N, **kw)
    ga.run_gather(ga.pointer_table(deltas), torch.tensor([x.data_ptr() for x in it], dtype=torch.int64).to(dev), ct2)
    torch.cuda.synchronize()
    m = _same(r3, ga)
    if m:
        msgs.append("gather: " + m)
    # both at once: gather through the index lists straight from fine-tuned + base tensors vs ingest + gather
    r4 = CompressPlan(sizes, N, **kw)
    ingv = [ing[p * N:(p + 1) * N] for p in range(P)]
    itab = torch.tensor([x.data_ptr() for x in it], dtype=torch.int64).to(dev)
    r4.run_gather(r4.pointer_table(ingv), itab, ct2)
    gb = CompressPlan(sizes, N, **kw)
    gb.run_gather_from_base(gb.pointer_table(fts), torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev),
                            itab, ct2)
    torch.cuda.synchronize()
    m = _same(r4, gb)
    if m:
        msgs.append("gather_from_base: " + m)
    # the consumers of a masked plan: merge with the mask scatter inside the streaming launch (svdq_merge_masked, + base)
    # against merging the compacted rows and scattering with torch's boolean assignment -- bit for bit; the masked
    # plan-level diagnostics against the plain ones on the compacted copies (fp64 sums in another order)
    mtab_i = torch.tensor([x.data_ptr() for x in ms._i["mb"]], dtype=torch.int64).to(dev)
    us_g = ms.unit_starts(ga, ct2, mask_table=mtab_i)
    wts = torch.rand(1, N, device=dev, generator=g) + 0.1
    wts = (wts / wts.sum()).contiguous()
    full = [torch.empty(D, device=dev) for D in sizes]
    otab = torch.tensor([f.data_ptr() for f in full], dtype=torch.int64).to(dev)
    btab_c = torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev)
    ga.merge_masked(wts, mtab_i, us_g, ct2, otab, fill=torch.ones(P, dtype=torch.int32, device=dev), base_table=btab_c)
    cbuf, coffs = r3.merge(wts, rows_dev=ct)
    torch.cuda.synchronize()
    counts = ct.cpu().tolist()
    for p in range(P):
        want = torch.zeros(sizes[p], device=dev)
        want[masks[p]] = cbuf[coffs[p]:coffs[p] + counts[p]]
        want = base[p] + want
        if not torch.equal(torch.nan_to_num(full[p], nan=7.0), torch.nan_to_num(want, nan=7.0)):
            msgs.append(f"merge_masked: parameter {p} differs")
            break
    dm = ga.diagnostics_masked(ga.pointer_table(deltas), mtab_i, us_g, ct2).cpu().numpy()
    dc = r3.diagnostics(r3.pointer_table(dt), ct).cpu().numpy()
    ok = np.isclose(dm, dc, rtol=1e-6, atol=1e-12) | (np.isnan(dm) & np.isnan(dc))
    if not ok.all():
        p_, t_, j_ = [int(v[0]) for v in np.nonzero(~ok)]
        msgs.append(f"diagnostics_masked: parameter {p_} task {t_} field {j_}: {dm[p_, t_, j_]} vs {dc[p_, t_, j_]}")
    extra = []
    mtab = torch.tensor([x.data_ptr() for x in ms._i["mb"]], dtype=torch.int64).to(dev)
    if N > 16:
        # the mask walk above 16 tasks (one-wave kernels) against the compacted run on the two-wave kernels: N = 17..20
        # default to the 4x4-block pass 2, whose task sums associate differently -- plan flag 8 selects the two-wave one
        rw = r3
        if N <= 20:
            rw = CompressPlan(sizes, N, flags=8, **kw)
            rw.run(rw.pointer_table(dt), ct)
            extra.append(rw)
        wk = CompressPlan(sizes, N, **kw)
        wk.run_masked(wk.pointer_table(deltas), mtab, ms.unit_starts(wk, ct2, mask_table=mtab), ct2)
        torch.cuda.synchronize()
        m = _same(rw, wk)
        if m:
            msgs.append("walk (one-wave kernels): " + m)
        extra.append(wk)
    if N <= 16:
        # the mask walk (no index lists): against the compacted run, and straight from checkpoints against ingest + walk
        wk = CompressPlan(sizes, N, **kw)
        us = ms.unit_starts(wk, ct2, mask_table=mtab)
        wk.run_masked(wk.pointer_table(deltas), mtab, us, ct2)
        torch.cuda.synchronize()
        m = _same(r3, wk)
        if m:
            msgs.append("walk: " + m)
        w4 = CompressPlan(sizes, N, **kw)
        w4.run_masked(w4.pointer_table(ingv), mtab, ms.unit_starts(w4, ct2, mask_table=mtab), ct2)
        wb = CompressPlan(sizes, N, **kw)
        wb.run_masked_from_base(wb.pointer_table(fts), torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev),
                                mtab, ms.unit_starts(wb, ct2, mask_table=mtab), ct2)
        torch.cuda.synchronize()
        m = _same(w4, wb)
        if m:
            msgs.append("walk_from_base: " + m)
        extra += [wk, w4, wb]
    for pl in [r2, fb, r3, ga, r4, gb] + extra:
        pl.close()
    return desc, msgs


def diag_case(sq, orc, dev, seed: int, c: int):
    """All six numbers of every (parameter, task) of a random plan against the reference's formula in fp64 on the plan's
    own artifacts (helpers.diag_check: 2e-6 + the fp32 forward-error bound of the formula), for the plain kernel, the
    add_mean extension and the masked walk in one polarity."""
    from helpers import diag_check
    from svdq_amd.pipeline import CompressPlan
    from svdq_amd.mask_loader import MaskSet
    from svdq_amd import diagnostics as dg
    rnd = random.Random(9000011 * seed + c)
    g = torch.Generator().manual_seed(131 * seed + c)
    N = rnd.choice([1, 2, 3, 4, 5, 7, 8, 8, 9, 12, 15, 16, 17, 19, 20, 20, 21, 24, 25, 28, 31, 32])
    P = rnd.randint(1, 4)
    sizes = [rnd.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4095, 4096, 4097, 8193,
                         rnd.randint(1, 40000)]) for _ in range(P)]
    fp16, center = rnd.random() < 0.7, rnd.random() < 0.5
    bits, stages = rnd.choice([4, 8]), rnd.choice([1, 2, 3])
    unit_rows = rnd.choice([0, 0, 1024, 4096])
    dens = rnd.choice([0.03, 0.3, 0.6, 0.94, 1.0])
    inverted = rnd.random() < 0.4
    desc = (f"N={N} sizes={sizes} fp16={fp16} center={center} bits={bits} stages={stages} unit_rows={unit_rows} "
            f"dens={dens} inverted={inverted}")
    kw = dict(energy_threshold=rnd.choice([0.5, 0.9, 0.99]), max_rank=rnd.choice([None, 2, 64]), center=center, fp16=fp16,
              low_bits=bits, rtvq_stages=stages, device=dev, unit_rows=unit_rows)
    vecs = []
    for i, D in enumerate(sizes):
        ds = [d.clone() for d in orc.synthetic_deltas(D, N, 40000 + 97 * seed + 13 * c + i, rank=min(3, N))]
        for t, d in enumerate(ds):      # a spike somewhere: a dropped or doubled row cannot hide in the sums
            d[rnd.randrange(D)] += 2.0 + 0.1 * t
        vecs.append([d.to(dev) for d in ds])
    msgs = []

    def artifacts(plan, sm, p, t):
        k, r, rows = int(sm.k[p]), int(sm.r[p]), int(sm.rows[p])
        Uh, Ul, mean = plan.basis_tensors(p, k, r, rows)
        ch = torch.from_numpy(sm.c_high[p, t, :k].astype(np.float32))
        nl = r - k
        cl = torch.zeros(0)
        if nl > 0:
            cl = torch.from_numpy(orc.rtvq_dequantize({"codes": sm.codes[p, t, :, :nl], "scale": sm.scale[p, t],
                                                       "zero_point": sm.zero_point[p, t]}).reshape(-1).copy())
        return Uh.cpu(), Ul.cpu(), ch, cl, (mean.cpu() if mean is not None else None)

    def check(tag, plan, sm, res, xs, with_mean=False):
        for p in range(P):
            if int(sm.rows[p]) <= 0:
                continue
            for t in range(N):
                Uh, Ul, ch, cl, mean = artifacts(plan, sm, p, t)
                if not (torch.isfinite(cl).all() and np.isfinite(res[p, t]).all()):
                    if torch.isfinite(cl).all() and int(sm.r[p]) - int(sm.k[p]) > 2:
                        msgs.append(f"{tag}: parameter {p} task {t}: non-finite numbers from finite artifacts")
                        return
                    continue      # the degenerate quantizer's NaN, as in the reference
                try:
                    diag_check(dict(zip(orc.DIAG_KEYS, res[p, t])), xs[p][t], Uh, Ul, ch, cl, what=(tag, p, t),
                               mean=mean if with_mean else None)
                except AssertionError as e:
                    msgs.append(str(e)[:300])
                    return

    plan = CompressPlan(sizes, N, **kw)
    table = plan.pointer_table(vecs)
    plan.run(table)
    sm = plan.fetch_small()
    xs = [[v.cpu() for v in vs] for vs in vecs]
    check("k_diag", plan, sm, plan.diagnostics(table).cpu().numpy(), xs)
    if center and not msgs:
        check("add_mean", plan, sm, plan.diagnostics(table, add_mean=True).cpu().numpy(), xs, with_mean=True)
    if not msgs:      # the per-call kernel on one (parameter, task)
        p, t = rnd.randrange(P), rnd.randrange(N)
        Uh, Ul, ch, cl, _ = artifacts(plan, sm, p, t)
        if torch.isfinite(cl).all():
            got = dg._fused_error(vecs[p][t], Uh.to(dev), Ul.to(dev), ch, cl, dev)
            if all(np.isfinite(v) for v in got.values()):
                try:
                    diag_check(got, xs[p][t], Uh, Ul, ch, cl, what=("recon_error", p, t))
                except AssertionError as e:
                    msgs.append(str(e)[:300])
    plan.close()
    if not msgs:      # the masked walk: compress compacted copies declared at the full size, diagnose through the mask
        masks = [(torch.rand(D, generator=g) < dens) for D in sizes]
        sel = [(~m if inverted else m) for m in masks]
        ms = MaskSet(sizes, dev)
        ct, cf = ms.count_scan([m.to(dev) for m in masks])
        rows_dev = cf if inverted else ct
        pm = CompressPlan(sizes, N, **dict(kw, center=False))
        mtab = torch.tensor([x.data_ptr() for x in ms._s["mb"]], dtype=torch.int64).to(dev)
        us = ms.unit_starts(pm, rows_dev, entry_map=[(q, inverted) for q in range(P)])
        comp = [[torch.cat([v[s_.to(dev)], torch.zeros(D - int(s_.sum()), device=dev)]) for v in vs]
                for vs, s_, D in zip(vecs, sel, sizes)]
        pm.run(pm.pointer_table(comp), rows_dev)
        smm = pm.fetch_small()
        res = pm.diagnostics_masked(table, mtab, us, rows_dev).cpu().numpy()
        check("walk", pm, smm, res, [[v.cpu()[s_] for v in vs] for vs, s_ in zip(vecs, sel)])
        pm.close()
        ms.close()
    return desc, msgs


def merge_case(sq, orc, dev, seed: int, c: int):
    """out = base + sum_s share_s * ((U_high cbar_s,high + U_low cbar_s,low + mean) * scale), cbar_s = sum_t w_st c_t over the
    present tasks in sorted order (merge.py:61-194; apply_weights_to_tensors; apply_merged_deltas) -- in fp64 from the
    plan's own basis, fp16 c_high and oracle-dequantized c_low, against svdq_merge element by element.  Tolerance per
    element: (r + n_sets + 6) 2^-24 times the sum of the magnitudes that enter it (every fp32 rounding of the kernel's
    chain is relative to a partial sum bounded by that), which is what an fp32 evaluation is entitled to."""
    from svdq_amd.pipeline import CompressPlan
    rnd = random.Random(6000011 * seed + c)
    g = torch.Generator().manual_seed(151 * seed + c)
    N = rnd.choice([1, 2, 3, 5, 8, 8, 9, 12, 16, 17, 20, 24, 32])
    P = rnd.randint(1, 4)
    sizes = [rnd.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4097, rnd.randint(1, 30000)]) for _ in range(P)]
    fp16, center = rnd.random() < 0.7, rnd.random() < 0.7
    bits, stages = rnd.choice([4, 8]), rnd.choice([1, 2])
    n_sets = rnd.choice([1, 1, 2, 3, 4, 8]) if N >= 2 else 1
    with_base, with_scale = rnd.random() < 0.6, rnd.random() < 0.4
    desc = f"N={N} sizes={sizes} fp16={fp16} center={center} bits={bits} stages={stages} sets={n_sets} base={with_base}"
    plan = CompressPlan(sizes, N, energy_threshold=rnd.choice([0.5, 0.9, 0.99]), max_rank=rnd.choice([None, 2, 64]),
                        center=center, fp16=fp16, low_bits=bits, rtvq_stages=stages, device=dev,
                        unit_rows=rnd.choice([0, 1024, 4096]))
    vecs = [[d.to(dev) for d in orc.synthetic_deltas(D, N, 70000 + 89 * seed + 17 * c + i, rank=min(3, N))]
            for i, D in enumerate(sizes)]
    plan.run(plan.pointer_table(vecs))
    sm = plan.fetch_small()
    # weights [P, S, N]: every task in exactly one set (or absent), renormalised inside the set like merge.py:123-124
    w = np.full((P, n_sets, N), -1.0, dtype=np.float32)
    for p in range(P):
        for t in range(N):
            if rnd.random() < 0.9 or t == 0:
                w[p, rnd.randrange(n_sets), t] = np.float32(0.05 + rnd.random())
        for s_ in range(n_sets):
            on = w[p, s_] >= 0
            if on.any():
                w[p, s_, on] = (w[p, s_, on] / w[p, s_, on].sum(dtype=np.float32)).astype(np.float32)
    share = None
    if n_sets > 1:
        raw = torch.softmax(torch.rand(n_sets, generator=g), 0).numpy().astype(np.float32)
        share = np.where((w >= 0).any(axis=2), raw[None, :], np.float32(-1.0)).astype(np.float32)      # [P, S]
    scale = (0.25 + torch.rand(P, generator=g)).numpy().astype(np.float32) if with_scale else None
    base = [torch.randn(D, generator=g) for D in sizes] if with_base else None
    buf, offs, otab = plan.new_merged_outputs()
    bd = [b.to(dev) for b in base] if base is not None else None
    plan.merge(torch.from_numpy(w).to(dev), set_share=torch.from_numpy(share).to(dev) if share is not None else None,
               scale=torch.from_numpy(scale).to(dev) if scale is not None else None,
               base_table=(torch.tensor([b.data_ptr() for b in bd], dtype=torch.int64).to(dev) if bd is not None else None),
               out_table=otab)
    torch.cuda.synchronize()
    msgs = []
    for p, D in enumerate(sizes):
        k, r = int(sm.k[p]), int(sm.r[p])
        Uh, Ul, mean = plan.basis_tensors(p, k, r, D)
        U = torch.cat([Uh, Ul], dim=1).double().cpu()
        cs = []
        for t in range(N):
            ch = sm.c_high[p, t, :k].astype(np.float64)
            cl = orc.rtvq_dequantize({"codes": sm.codes[p, t, :, :r - k], "scale": sm.scale[p, t],
                                      "zero_point": sm.zero_point[p, t]}).reshape(-1).astype(np.float64) if r > k else np.zeros(0)
            cs.append(np.concatenate([ch, cl]))
        cs = np.stack(cs)                                            # [N, r]
        if not np.isfinite(cs).all():
            continue                                                 # the degenerate quantizer's NaN
        mu = mean.flatten().double().cpu().numpy() if mean is not None else np.zeros(D)
        sc = float(scale[p]) if scale is not None else 1.0
        want, mag = np.zeros(D), np.zeros(D)
        Ua = U.numpy()
        for s_ in range(n_sets):
            on = w[p, s_] >= 0
            if not on.any():
                continue
            cbar = (w[p, s_, on].astype(np.float64)[:, None] * cs[on]).sum(0)
            cabs = (w[p, s_, on].astype(np.float64)[:, None] * np.abs(cs[on])).sum(0)
            sh = float(share[p, s_]) if share is not None else 1.0
            want += sh * (Ua @ cbar + mu) * sc
            mag += abs(sh) * (np.abs(Ua) @ cabs + np.abs(mu)) * abs(sc)
        if base is not None:
            want += base[p].double().numpy()
            mag += np.abs(base[p].double().numpy())
        got = buf[offs[p]:offs[p] + D].double().cpu().numpy()
        tol = (r + n_sets + 6) * 2.0 ** -24 * mag + 1e-30
        if not np.all(np.abs(got - want) <= tol):
            i = int(np.argmax(np.abs(got - want) / tol))
            msgs.append(f"merge: parameter {p} row {i}: {got[i]} vs {want[i]} (tol {tol[i]:.2e})")
            break
    plan.close()
    return desc, msgs
