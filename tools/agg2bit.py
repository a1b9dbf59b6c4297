import os, sys, random
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import svdq_amd as sq
from oracle import svd_hybrid_oracle as orc
dev = torch.device("cuda", 0)
rnd = random.Random(11)
ratios = []
for c in range(60):
    N = rnd.choice([6, 8, 11, 16, 20])
    D = rnd.choice([1000, 4099, 30957, 152516])
    thr = rnd.choice([0.9, 0.95]); S = rnd.choice([1, 2, 3]); center = rnd.random() < 0.7
    deltas = orc.synthetic_deltas(D, N, 9000 + c, rank=3)
    ref = orc.compress_parameter(deltas, thr, 64, center, True, 2, S)
    plan, sm = sq.compress_batch([[d.to(dev) for d in deltas]], energy_threshold=thr, max_rank=64, center=center, fp16=True,
                                 low_bits=2, rtvq_stages=S, device=dev)
    k, r = int(sm.k[0]), int(sm.r[0])
    if k != ref["basis"]["k"] or r - k <= 2: continue
    U_high, U_low, mean = plan.basis_tensors(0, k, r, D)
    quant = sq.RTVQQuantizer(2, S)
    eo = er = 0.0
    for t in range(N):
        art = sq.pipeline.task_artifact(plan, sm, 0, t)
        cl = quant.dequantize(art["c_low_quant"], device=dev).float()
        rec = sq.reconstruct_from_coefficients(art["c_high_fp16"].to(dev).float(), cl, U_high, U_low, dev, mean=mean).cpu().numpy()
        x = deltas[t].numpy(); rr = ref["recon"][t].numpy()
        eo += np.linalg.norm(rec - x) ** 2; er += np.linalg.norm(rr - x) ** 2
    ratios.append((eo / er) ** 0.5)
    print(f"N={N} D={D} S={S} center={center}: rms error ours/ref = {ratios[-1]:.3f}", flush=True)
ratios = np.array(ratios)
print("cases", len(ratios), "geomean", float(np.exp(np.log(ratios).mean())), "min", ratios.min(), "max", ratios.max())
