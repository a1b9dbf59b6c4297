"""
Synthetic workloads of BASELINE.json: CLIP ViT visual-encoder state-dict shapes (SURVEY.md section 8,
derived from open_clip's VisionTransformer as used by the reference notebook) and the on-device
generator of decaying-spectrum task deltas that keeps the quantizer input
non-degenerate.  No checkpoints or datasets exist offline; shapes are what matters.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

VIT_SPECS = {
    #            width layers patch tokens out
    "ViT-B-32": (768, 12, 32, 50, 512),
    "ViT-B-16": (768, 12, 16, 197, 512),
    "ViT-L-14": (1024, 24, 14, 257, 768),
}


def vit_visual_shapes(model: str) -> Dict[str, Tuple[int, ...]]:
    """name -> shape of ``model.visual.state_dict()`` for the CLIP visual tower."""
    w, L, patch, tokens, out = VIT_SPECS[model]
    s: Dict[str, Tuple[int, ...]] = {
        "class_embedding": (w,), "positional_embedding": (tokens, w), "proj": (w, out),
        "conv1.weight": (w, 3, patch, patch), "ln_pre.weight": (w,), "ln_pre.bias": (w,),
        "ln_post.weight": (w,), "ln_post.bias": (w,),
    }
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        s.update({
            p + "ln_1.weight": (w,), p + "ln_1.bias": (w,),
            p + "attn.in_proj_weight": (3 * w, w), p + "attn.in_proj_bias": (3 * w,),
            p + "attn.out_proj.weight": (w, w), p + "attn.out_proj.bias": (w,),
            p + "ln_2.weight": (w,), p + "ln_2.bias": (w,),
            p + "mlp.c_fc.weight": (4 * w, w), p + "mlp.c_fc.bias": (4 * w,),
            p + "mlp.c_proj.weight": (w, 4 * w), p + "mlp.c_proj.bias": (w,),
        })
    return s


def numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


def synth_task_buffers(rows: List[int], n_tasks: int, seed: int, device, rank: int = 3, a: float = 0.01,
                       eps: float = 0.002) -> Tuple[List[torch.Tensor], List[List[torch.Tensor]]]:
    """One flat fp32 buffer per task holding every parameter (64-float aligned offsets), filled on
    the device with delta_t = a * B (g_t * s) + eps * n_t, s = (1, .5, .25, ...).
    Returns (task buffers, views[p][t])."""
    dev = torch.device(device)
    offs, tot = [], 0
    for d in rows:
        offs.append(tot)
        tot += (d + 63) // 64 * 64
    g = torch.Generator(device=dev).manual_seed(seed)
    bufs = [torch.empty(tot, dtype=torch.float32, device=dev) for _ in range(n_tasks)]
    s = torch.tensor([0.5 ** i for i in range(rank)], device=dev)
    views: List[List[torch.Tensor]] = []
    for d, o in zip(rows, offs):
        B = torch.randn(d, rank, device=dev, generator=g)
        vs = []
        for t in range(n_tasks):
            gt = torch.randn(rank, device=dev, generator=g)
            v = bufs[t][o:o + d]
            torch.randn(d, device=dev, generator=g, out=v)
            v.mul_(eps).add_((B * (gt * s)).sum(dim=1), alpha=a)
            vs.append(v)
        views.append(vs)
        del B
    return bufs, views
