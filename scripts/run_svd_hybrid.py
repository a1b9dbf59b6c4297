#!/usr/bin/env python3
"""SVD-Hybrid merging on an MI355X, driven by the reference's command-line flags.

Equivalent of the reference's scripts/run_svd_hybrid.py: everything after the flags is handled by
svdq_amd.cli (checkpoints -> task vectors -> masks -> bases + compression -> weights -> merge -> diagnostics ->
artifacts), every stage on the HIP kernels of this repository.
"""
from pathlib import Path
import sys

REPO_ROOT = Path(__file__).resolve().parents[1]
if str(REPO_ROOT) not in sys.path:
    sys.path.insert(0, str(REPO_ROOT))


def _run(argv=None):
    import svdq_amd
    return svdq_amd.cli.main(argv)


if __name__ == "__main__":
    _run()
