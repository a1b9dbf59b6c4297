"""Would hiding the eigen-stage behind the streaming passes pay?  The parameters are cut into two halves;
eig(A) runs on a side stream while pass 1 of B streams, eig(B) while pass 2 of A streams (four event hops).
Measured with the *_range entry points from Python against the plain sequence, for a whole model and for one rank's
share of the 8-GPU run.   usage: overlap_probe.py [N] [world]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdq_amd import workloads, shard
from svdq_amd.pipeline import CompressPlan

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shapes = workloads.vit_visual_shapes("ViT-L-14")
rows_all = [workloads.numel(shapes[n]) for n in sorted(shapes)]
rows = [rows_all[i] for i in shard.partition_lpt(rows_all, world)[0]] if world > 1 else rows_all
bufs, views = workloads.synth_task_buffers(rows, N, seed=1, device=dev)
plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev)
table = plan.pointer_table(views)
plan.tune_placement(table)
P = len(rows)
half, acc = 0, 0
for p, d in enumerate(rows):
    acc += d
    if acc >= sum(rows) / 2:
        half = p + 1
        break
s = torch.cuda.current_stream()
s2 = torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(4)]


def plain():
    plan.run(table)


def overlapped():
    e1, e2, e3, e4 = evs
    plan.gram_range(table, 0, half, s); e1.record(s)
    plan.gram_range(table, half, P - half, s); e3.record(s)
    s2.wait_event(e1); plan.eig_range(table, 0, half, s2); e2.record(s2)
    s2.wait_event(e3); plan.eig_range(table, half, P - half, s2); e4.record(s2)
    s.wait_event(e2); plan.bp_range(table, 0, half, s)
    s.wait_event(e4); plan.bp_range(table, half, P - half, s)
    plan.coeff_range(0, P, s)


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


plain(); torch.cuda.synchronize(); ref = plan.small.clone()
overlapped(); torch.cuda.synchronize()
same = torch.equal(plan.small, ref)
for _ in range(2):
    print(f"N={N} world={world} ({P} tensors, split at {half}): plain {ev(plain):.4f} ms, two halves overlapped {ev(overlapped):.4f} ms, "
          f"same small artifacts: {same}", flush=True)
