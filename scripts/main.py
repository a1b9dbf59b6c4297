#!/usr/bin/env python3
"""Dispatcher with the reference's ``src/main.py`` command line (main.py:41-89): ``--method svd_hybrid`` (the default,
and the only method the reference implements) hands every remaining flag to the SVD-Hybrid command line of this
repository; the three names the reference lists but does not implement answer exactly as it does.

    python scripts/main.py --method svd_hybrid --tasks Cars DTD ... --checkpoint-dir ... --base-model-path ...
"""
from pathlib import Path
import argparse
import sys

REPO_ROOT = Path(__file__).resolve().parents[1]
if str(REPO_ROOT) not in sys.path:
    sys.path.insert(0, str(REPO_ROOT))


def main(argv=None):
    parser = argparse.ArgumentParser(description="Task merging methods for multi-task models")
    parser.add_argument("--method", type=str, default="svd_hybrid",
                        choices=["svd_hybrid", "task_arithmetic", "ties", "dare"],
                        help="Merging method to use (default: svd_hybrid)")
    args, remaining = parser.parse_known_args(argv)
    if args.method == "svd_hybrid":
        import svdq_amd
        return svdq_amd.cli.main(remaining)
    print(f"Method '{args.method}' not yet implemented")
    print(f"Use svd_hybrid for now, or implement {args.method} in a new module")
    return None


if __name__ == "__main__":
    main()
