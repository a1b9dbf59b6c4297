"""Long randomised parity run against the CPU oracle: python tools/fuzz_oracle.py [seed] [cases].
The cases are tests/fuzz_cases.oracle_case (the GPU test-suite runs a seeded subset of the same generator)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import svdq_amd as sq
from oracle import svd_hybrid_oracle as orc
from fuzz_cases import oracle_case

dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for c in range(cases):
    desc, msgs = oracle_case(sq, orc, dev, seed, c)
    bad += bool(msgs)
    print(f"case {c:3d}: {desc}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}", flush=True)
print(f"{cases - bad} / {cases} cases within tolerance", flush=True)
sys.exit(1 if bad else 0)
