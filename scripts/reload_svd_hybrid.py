#!/usr/bin/env python3
"""Rebuild a merged model from stored artifacts and (optionally) verify it against the saved merged model
(reference scripts/reload_svd_hybrid.py; reload.py:142-238).  The original fine-tuned checkpoints are not needed."""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from svdq_amd.storage import reconstruct_from_artifacts  # noqa: E402


def compute_state_dict_checksum(state_dict) -> str:
    h = hashlib.md5()
    for key in sorted(state_dict.keys()):
        if isinstance(state_dict[key], torch.Tensor):
            h.update(state_dict[key].cpu().numpy().tobytes())
    return h.hexdigest()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Reload an SVD-Hybrid merged model from artifacts")
    ap.add_argument("--artifact-dir", required=True)
    ap.add_argument("--base-model-path", required=True)
    ap.add_argument("--output-path", default=None)
    ap.add_argument("--verify", default=None, help="path of a saved merged_state_dict.pt to compare with")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    res = reconstruct_from_artifacts(args.artifact_dir, args.base_model_path, args.output_path, device=args.device)
    merged = res["merged_state_dict"]
    print(f"reloaded {len(merged)} entries, checksum {compute_state_dict_checksum(merged)}")
    ok = True
    if args.verify:
        saved = torch.load(args.verify, map_location="cpu", weights_only=True)
        worst = 0.0
        for k, v in saved.items():
            if isinstance(v, torch.Tensor) and v.is_floating_point():
                worst = max(worst, float((merged[k].cpu().float() - v.float()).abs().max()))
        ok = worst <= 1e-5
        print(f"max |reloaded - saved| = {worst:.3e} -> {'MATCH' if ok else 'MISMATCH'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
