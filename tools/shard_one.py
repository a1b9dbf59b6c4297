"""One rank's share of the 8-way strong-scaling run (config #4) on a single GPU: LPT shard 0 of ViT-L-14 x 8, step time
against the rows per work unit (a shard has ~4 700 units of 8 192 rows for 5 120 resident waves: one partial wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdq_amd import workloads, shard
from svdq_amd.pipeline import CompressPlan

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shapes = workloads.vit_visual_shapes("ViT-L-14")
names = sorted(shapes)
rows_all = [workloads.numel(shapes[n]) for n in names]
mine = shard.partition_lpt(rows_all, world)[0]
rows = [rows_all[i] for i in mine]
bufs, views = workloads.synth_task_buffers(rows, N, seed=1, device=dev)
print(f"shard 0 of {world}: {len(rows)} tensors, sum D = {sum(rows)} ({sum(rows) * N * 4 / 1e9:.2f} GB of deltas)", flush=True)
for ur in (0, 4096, 2048, 1024, 512):
    plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev, unit_rows=ur)
    table = plan.pointer_table(views)
    for _ in range(5):
        plan.run(table)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    tot = [0.0] * 4
    reps = 20
    for _ in range(reps):
        ev[0].record(); plan.gram_center(table)
        ev[1].record(); plan.eig_rank_select(table)
        ev[2].record(); plan.basis_project(table)
        ev[3].record(); plan.coeff_quantize()
        ev[4].record()
        torch.cuda.synchronize()
        for i in range(4):
            tot[i] += ev[i].elapsed_time(ev[i + 1])
    t = [x / reps for x in tot]
    print(f"unit_rows {ur or 'auto(8192)':>10}: units {plan.sizes.n_units:6d}  gram {t[0]:.3f}  eig {t[1]:.3f}  "
          f"basis_project {t[2]:.3f}  coeff {t[3]:.3f}  sum {sum(t):.3f} ms", flush=True)
    plan.close()

# the same step as one HIP graph launch (six kernels + a memset captured once): does it close the launch gaps?
plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev)
table = plan.pointer_table(views)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        plan.run(table)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        plan.run(table)
torch.cuda.synchronize()
for name, fn in (("eager", lambda: plan.run(table)), ("graph replay", g.replay)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50:.4f} ms per step (50 steps back to back)", flush=True)
