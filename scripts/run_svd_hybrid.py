#!/usr/bin/env python3
"""Run SVD-Hybrid merging on an MI355X with the reference's command line (reference scripts/run_svd_hybrid.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from svdq_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
