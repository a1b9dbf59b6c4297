"""Long randomised run of the plan-level merge (svdq_merge) against merge.py:61-194, 429-552 in fp64 on the plan's own artifacts:
python tools/fuzz_merge.py [seed] [cases].  The cases are tests/fuzz_cases.merge_case (the GPU suite runs a seeded subset)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import svdq_amd as sq
from oracle import svd_hybrid_oracle as orc
from fuzz_cases import merge_case

dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for c in range(cases):
    desc, msgs = merge_case(sq, orc, dev, seed, c)
    bad += bool(msgs)
    print(f"case {c:3d}: {desc}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}", flush=True)
print(f"{cases - bad} / {cases} cases within tolerance", flush=True)
sys.exit(1 if bad else 0)
