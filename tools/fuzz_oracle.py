"""Randomised parity against the CPU oracle (the restated reference op sequence): random (D, N, thresholds, centre,
fp16, bits, stages); compares singular values, rank (when the energy threshold is not within 1e-4 of a step of the
cumulative spectrum), retained energy and reconstructions (MSE <= 1e-6 where the reference is finite)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import svdq_amd as sq
from oracle import svd_hybrid_oracle as orc

dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rnd = random.Random(seed)
bad = 0
for c in range(cases):
    N = rnd.choice([1, 2, 3, 4, 6, 8, 8, 11, 16, 17, 20, 27, 32])
    D = rnd.choice([1, 2, 5, 31, 255, 256, 257, 1000, 4099, rnd.randint(1, 60000), rnd.randint(1, 200000)])
    thr = rnd.choice([0.5, 0.9, 0.95, 0.999, 1.0])
    max_rank = rnd.choice([None, 1, 3, 64])
    center, fp16 = rnd.random() < 0.75, rnd.random() < 0.75
    bits, stages = rnd.choice([2, 4, 8]), rnd.choice([1, 2, 3])
    deltas = orc.synthetic_deltas(D, N, 5000 + 31 * seed + c, rank=min(3, N))
    ref = orc.compress_parameter(deltas, thr, max_rank, center, fp16, bits, stages)
    plan, sm = sq.compress_batch([[d.to(dev) for d in deltas]], energy_threshold=thr, max_rank=max_rank, center=center,
                                 fp16=fp16, low_bits=bits, rtvq_stages=stages, device=dev)
    k, r = int(sm.k[0]), int(sm.r[0])
    msgs = []
    eo2 = er2 = 0.0
    S_ref = ref["basis"]["singular_values"].numpy()
    if r != len(S_ref):
        msgs.append(f"r {r} vs {len(S_ref)}")
    else:
        real = S_ref > 1e-3 * max(S_ref[0], 1e-30)   # the fp32-product Gram resolves sigma to ~1e-4 sigma_0
        if not np.allclose(sm.sigma[0, :r][real], S_ref[real], rtol=1e-4):
            msgs.append("sigma")
        e = S_ref.astype(np.float32) ** 2
        cum = np.cumsum(e, dtype=np.float32) / max(e.sum(dtype=np.float32), 1e-30)
        near = np.any(np.abs(cum - thr) < 1e-4)
        if not near and k != ref["basis"]["k"]:
            msgs.append(f"k {k} vs {ref['basis']['k']}")
        if k == ref["basis"]["k"] and r - k > 2:
            U_high, U_low, mean = plan.basis_tensors(0, k, r, D)
            quant = sq.RTVQQuantizer(bits, stages)
            for t in range(N):
                art = sq.pipeline.task_artifact(plan, sm, 0, t)
                cl = quant.dequantize(art["c_low_quant"], device=dev).float()
                rec = sq.reconstruct_from_coefficients(art["c_high_fp16"].to(dev).float(), cl, U_high, U_low, dev,
                                                       mean=mean).cpu().numpy()
                rr = ref["recon"][t].numpy()
                if np.isfinite(rr).all() and np.isfinite(rec).all():
                    mse = float(np.mean((rec - rr) ** 2))
                    if bits > 2 and mse > 1e-6:
                        msgs.append(f"recon mse {mse:.2e} task {t}")
                        break
                    if bits <= 2:   # the basis inside near-degenerate singular subspaces is not unique and 2-bit
                        x = deltas[t].numpy()       # noise is of the order of the bound: compare quality, aggregated
                        eo2 = eo2 + float(np.linalg.norm(rec - x) ** 2) if t else float(np.linalg.norm(rec - x) ** 2)
                        er2 = er2 + float(np.linalg.norm(rr - x) ** 2) if t else float(np.linalg.norm(rr - x) ** 2)
                        if t == N - 1 and eo2 > 2.5 ** 2 * er2 + 1e-12:
                            msgs.append(f"2-bit rms recon error {eo2 ** 0.5:.3e} vs reference {er2 ** 0.5:.3e}")
    bad += bool(msgs)
    print(f"case {c:3d}: D={D} N={N} thr={thr} max_rank={max_rank} center={center} fp16={fp16} b={bits} S={stages}: "
          f"{'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}", flush=True)
print(f"{cases - bad} / {cases} cases within tolerance", flush=True)
sys.exit(1 if bad else 0)
