"""Pass-2 time against how the output buffers are allocated: (A) basis and mean as two allocations (the default),
(B) one allocation holding both, (C) like B with the mean 64 MiB + 4 KiB past the basis.  Several candidates of each
in one process; the question is whether a layout removes the slow levels (2.8-3.05 ms against 2.72)."""
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
bb, mf = plan.sizes.basis_bytes, plan.sizes.mean_floats


def timed(reps=8):
    plan.basis_project(table); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.basis_project(table)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


keep = []
for trial in range(6):
    out = {}
    # A: two allocations
    plan.basis = torch.empty(bb, dtype=torch.uint8, device=dev)
    plan.mean = torch.empty(mf, dtype=torch.float32, device=dev)
    plan._typed = None
    keep += [plan.basis, plan.mean]
    out["A two allocs"] = timed()
    # B: one allocation, mean right behind the basis
    gap = (bb + 255) // 256 * 256
    one = torch.empty(gap + mf * 4, dtype=torch.uint8, device=dev)
    keep.append(one)
    plan.basis = one[:bb]
    plan.mean = one[gap:gap + mf * 4].view(torch.float32)
    out["B one alloc"] = timed()
    # C: one allocation, mean 64 MiB + 4 KiB further
    gap2 = gap + (64 << 20) + 4096
    one2 = torch.empty(gap2 + mf * 4, dtype=torch.uint8, device=dev)
    keep.append(one2)
    plan.basis = one2[:bb]
    plan.mean = one2[gap2:gap2 + mf * 4].view(torch.float32)
    out["C one alloc, shifted mean"] = timed()
    # D: mean in FRONT of the basis
    front = (mf * 4 + 255) // 256 * 256
    one3 = torch.empty(front + bb, dtype=torch.uint8, device=dev)
    keep.append(one3)
    plan.mean = one3[:mf * 4].view(torch.float32)
    plan.basis = one3[front:front + bb]
    out["D mean first"] = timed()
    print(f"trial {trial}: " + "  ".join(f"{k} {v:.3f}" for k, v in out.items()), flush=True)
    if len(keep) > 12:
        del keep[:5]
