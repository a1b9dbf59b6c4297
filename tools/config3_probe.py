"""BASELINE.json configs[2]: ViT-B-16, 8 tasks, tall-mask union, 4-stage 4-bit RTVQ, 1 x MI355X.
Times the pieces of driver.build_bases as it stands (per-parameter compaction launches + one plan)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd
from svdq_amd import workloads
dev = torch.device("cuda", 0)
N = 8
shapes = workloads.vit_visual_shapes("ViT-B-16")
names = sorted(shapes)
rows = [workloads.numel(shapes[n]) for n in names]
bufs, views = workloads.synth_task_buffers(rows, N, seed=5, device=dev)
tasks = [f"T{i}" for i in range(N)]
tv = {t: {n: views[p][i].view(shapes[n]) for p, n in enumerate(names)} for i, t in enumerate(tasks)}
g = torch.Generator(device=dev).manual_seed(11)
task_masks = {t: {n: (torch.rand(shapes[n], device=dev, generator=g) > 0.7) for n in names} for t in tasks}
cfg = svdq_amd.SVDHybridConfig(svd_energy_threshold=0.9, svd_max_rank=64, svd_center=True, svd_fp16=True,
                               svd_low_bits=4, svd_rtvq_stages=4, svd_mask_strategy="union")
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out
t_comb, combined = timed(lambda: svdq_amd.combine_masks(task_masks, strategy="union", device="cuda", verbose=False))
dens = sum(float(m.sum()) for m in combined.values()) / sum(rows)
t_bases, bases = timed(lambda: svdq_amd.build_bases(tv, combined, cfg, "cuda"))
t_all, (b2, comp) = timed(lambda: svdq_amd.run_basis_and_compress(tv, combined, cfg, "cuda"), reps=1)
scal = sum(rows) * N
print(f"union density {dens:.3f}; combine_masks {t_comb*1e3:.1f} ms; build_bases (compaction + plan + 4 launches + D2H + dicts) "
      f"{t_bases*1e3:.1f} ms = {scal/t_bases/1e6:.0f} MParams/s; + compress_all_parameters dict assembly {t_all*1e3:.1f} ms")
b = bases["transformer.resblocks.0.mlp.c_fc.weight"]["masked"]
print("c_fc k", b["k"], "D", b["D"], "of", rows[names.index("transformer.resblocks.0.mlp.c_fc.weight")])
