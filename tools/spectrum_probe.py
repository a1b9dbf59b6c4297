"""Singular values of the graded-spectrum fixtures: HIP path (fp64-MFMA Gram, and the fp32-product Gram with
flags bit 1) against the reference's LAPACK values and the fp64 values stored beside them."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svdq_amd  # noqa: E402
from svdq_amd.pipeline import CompressPlan  # noqa: E402
from helpers import load_golden, as_tensors  # noqa: E402

dev = torch.device("cuda", 0)
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "spectrum_*.npz"))):
    g = load_golden(os.path.basename(path))
    vs = [d.to(dev) for d in as_tensors(g["deltas"])]
    N, D = int(g["N"]), int(g["D"])
    print(os.path.basename(path), "k_ref", int(g["k"]), "energy_ref", float(g["energy_retained"]))
    for flags in (0, 2):
        plan = CompressPlan([D], N, energy_threshold=float(g["thr"]), max_rank=None, center=bool(g["center"]),
                            fp16=True, low_bits=4, rtvq_stages=2, device=dev, flags=flags)
        plan.run(plan.pointer_table([vs]))
        sm = plan.fetch_small()
        S, S64, sig = g["S"], g["S_f64"], sm.sigma[0]
        U_high, U_low, _ = plan.basis_tensors(0, int(sm.k[0]), int(sm.r[0]), D)
        U = torch.cat([U_high, U_low], 1).double()
        norms = (U * U).sum(0).sqrt().cpu().numpy()
        print(f"  flags={flags} k={int(sm.k[0])} energy={float(sm.energy[0]):.8f}")
        for i in range(N):
            print(f"    s_ref {S[i]:.7e} ours {sig[i]:.7e} rel_vs_ref {abs(sig[i]-S[i])/max(S[i],1e-300):.1e} "
                  f"rel_vs_f64 {abs(sig[i]-S64[i])/max(S64[i],1e-300):.1e} ref_vs_f64 {abs(S[i]-S64[i])/max(S64[i],1e-300):.1e} "
                  f"|u|={norms[i]:.5f}")
