"""Long randomised cross-check of the input modes of svdq_compress: python tools/fuzz_modes.py [seed] [cases].
The cases are tests/fuzz_cases.modes_case (the GPU test-suite runs a seeded subset of the same generator)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import svdq_amd as sq
from fuzz_cases import modes_case

dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for c in range(cases):
    desc, msgs = modes_case(sq, dev, seed, c)
    bad += bool(msgs)
    print(f"case {c:3d}: {desc}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}", flush=True)
print(f"{cases - bad} / {cases} cases identical", flush=True)
sys.exit(1 if bad else 0)
