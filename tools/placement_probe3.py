"""Pass-2 time against WHERE in device memory the output allocation lands: candidates are allocated one after the
other while the earlier ones stay held, so they walk through the 288 GB; for each the virtual address, the
distance (in allocation order) from the inputs and pass-2 time are printed.  Question: is the level a function of the
region of memory (e.g. the stack layer / rank the pages belong to), i.e. predictable, or not?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdq_amd import workloads
from svdq_amd.pipeline import CompressPlan

dev = torch.device("cuda", 0)
N = 8
shapes = workloads.vit_visual_shapes("ViT-L-14")
rows = [workloads.numel(shapes[k]) for k in sorted(shapes)]
bufs, views = workloads.synth_task_buffers(rows, N, seed=1, device=dev)
plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev)
table = plan.pointer_table(views)
plan.run(table); torch.cuda.synchronize()
in_lo = min(b.data_ptr() for b in bufs)
print(f"inputs: {len(bufs)} buffers from VA {in_lo:#x}; first output VA {plan.basis.data_ptr():#x}", flush=True)


def timed(reps=6):
    plan.basis_project(table); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.basis_project(table)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def timed_gram(reps=4):
    plan.gram_center(table); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.gram_center(table)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"pass 1 {timed_gram():.3f} ms; pass 2 into the plan's own allocation {timed():.3f} ms", flush=True)
keep = []
free0, total = torch.cuda.mem_get_info(dev)
n_cand = int((free0 - (12 << 30)) // (plan.sizes.basis_bytes + plan.sizes.mean_floats * 4 + (1 << 21)))
n_cand = min(n_cand, 40)
print(f"free {free0 / 2**30:.1f} GiB of {total / 2**30:.1f}; {n_cand} candidates of "
      f"{(plan.sizes.basis_bytes + plan.sizes.mean_floats * 4) / 2**30:.2f} GiB", flush=True)
for c in range(n_cand):
    b, m = plan._alloc_outputs()
    keep.append((b, m))
    plan.basis, plan.mean, plan._typed = b, m, None
    t = timed()
    free, _ = torch.cuda.mem_get_info(dev)
    print(f"cand {c:2d}  VA {b.data_ptr():#x}  (+{(b.data_ptr() - in_lo) / 2**30:7.2f} GiB from the inputs)  "
          f"used {(total - free) / 2**30:6.1f} GiB  pass 2 {t:.3f} ms", flush=True)
