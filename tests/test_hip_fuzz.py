"""Seeded subsets of the randomised parity generators (tests/fuzz_cases.py; tools/fuzz_*.py run them for longer):
40 random configurations against the CPU oracle, 20 against the alternative input modes 30 of the diagnostics kernels
and 30 of the plan-level merge against the reference's formulas in fp64, every run the same."""
import pytest
import torch

from fuzz_cases import oracle_case, modes_case, diag_case, merge_case

pytestmark = pytest.mark.gpu


def test_fuzz_oracle_seeded_40():
    import svdq_amd as sq
    from oracle import svd_hybrid_oracle as orc
    dev = torch.device("cuda", 0)
    bad = []
    for c in range(40):
        desc, msgs = oracle_case(sq, orc, dev, 0, c)
        if msgs:
            bad.append(f"case {c}: {desc}: {'; '.join(msgs)}")
    assert not bad, "\n".join(bad)


def test_fuzz_modes_seeded_20():
    import svdq_amd as sq
    dev = torch.device("cuda", 0)
    bad = []
    for c in range(20):
        desc, msgs = modes_case(sq, dev, 1, c)
        if msgs:
            bad.append(f"case {c}: {desc}: {'; '.join(msgs)}")
    assert not bad, "\n".join(bad)


def test_fuzz_diagnostics_seeded_30():
    import svdq_amd as sq
    from oracle import svd_hybrid_oracle as orc
    dev = torch.device("cuda", 0)
    bad = []
    for c in range(30):
        desc, msgs = diag_case(sq, orc, dev, 2, c)
        if msgs:
            bad.append(f"case {c}: {desc}: {'; '.join(msgs)}")
    assert not bad, "\n".join(bad)


def test_fuzz_merge_seeded_30():
    import svdq_amd as sq
    from oracle import svd_hybrid_oracle as orc
    dev = torch.device("cuda", 0)
    bad = []
    for c in range(30):
        desc, msgs = merge_case(sq, orc, dev, 3, c)
        if msgs:
            bad.append(f"case {c}: {desc}: {'; '.join(msgs)}")
    assert not bad, "\n".join(bad)
