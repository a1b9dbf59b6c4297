#!/usr/bin/env python3
"""The reference's top-level ``load_and_merge.py`` (load_and_merge.py:20-125): rebuild the merged model from an artifact
directory and a base checkpoint alone and write it to ``--output-path``.  Same four flags; ``--device`` may also be
``auto`` (or omitted in a call of ``reconstruct_from_artifacts``), which resolves to the GPU -- the only place this
package computes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import svdq_amd  # noqa: E402


def reconstruct_from_artifacts(artifact_dir: str, base_model_path: str, output_path: str, device=None):
    device = "cuda" if device in (None, "auto") else device
    print(f"Using device: {device}")
    res = svdq_amd.reconstruct_from_artifacts(artifact_dir, base_model_path, output_path, device=device)
    print(f"Saved merged model ({len(res['merged_state_dict'])} entries) to {output_path}")
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description="Reconstruct merged model from SVD-Hybrid artifacts")
    ap.add_argument("--artifact-dir", type=str, required=True, help="Directory containing artifacts")
    ap.add_argument("--base-model-path", type=str, required=True, help="Path to base model checkpoint")
    ap.add_argument("--output-path", type=str, required=True, help="Path to save reconstructed merged model")
    ap.add_argument("--device", type=str, default="cpu", help="cuda, cpu (results on the host) or auto")
    args = ap.parse_args(argv)
    reconstruct_from_artifacts(args.artifact_dir, args.base_model_path, args.output_path, args.device)
    return 0


if __name__ == "__main__":
    sys.exit(main())
