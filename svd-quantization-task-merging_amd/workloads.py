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
    the device with delta_t = a * B (g_t * s) + eps * n_t, s = (1, .5, .25, ...); B and g_t are drawn per
    parameter.  Returns (task buffers, views[p][t]).

    The whole batch is generated in O(n_tasks * rank) launches over the concatenated buffers (one ``normal_`` per
    task buffer, one per column of B, one gather + one fused multiply-add per task and column) -- not one handful of
    launches per (parameter, task): the ~14 000 tiny dispatches of the earlier per-tensor loop were the storm every
    ``rocprofv3 --pmc`` crash of round 2 died in (profiles/README.md), and they cost seconds of start-up.  The
    stream is synchronised before returning, so no generator work is in flight when the first svdq kernel starts."""
    dev = torch.device(device)
    offs, tot = [], 0
    for d in rows:
        offs.append(tot)
        tot += (d + 63) // 64 * 64
    g = torch.Generator(device=dev).manual_seed(seed)
    P = len(rows)
    sizes = torch.tensor([(d + 63) // 64 * 64 for d in rows], dtype=torch.int64, device=dev)
    pid = torch.repeat_interleave(torch.arange(P, dtype=torch.int32, device=dev), sizes, output_size=tot)
    s = torch.tensor([0.5 ** i for i in range(rank)], device=dev)
    gts = torch.randn(n_tasks, rank, P, device=dev, generator=g) * (a * s).view(1, rank, 1)   # a * g_t * s per parameter
    Bs = [torch.randn(tot, device=dev, generator=g) for _ in range(rank)]
    bufs = []
    for t in range(n_tasks):
        b = torch.empty(tot, dtype=torch.float32, device=dev)
        b.normal_(generator=g).mul_(eps)
        for r in range(rank):
            b.addcmul_(Bs[r], gts[t, r].index_select(0, pid))
        bufs.append(b)
    del Bs, pid, gts
    views = [[bufs[t][o:o + d] for t in range(n_tasks)] for d, o in zip(rows, offs)]
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()      # the generator's temporaries go back to the driver: output placement is tuned later
    return bufs, views
