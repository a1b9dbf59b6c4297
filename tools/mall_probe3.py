"""Does bp(A) still benefit from the Infinity Cache when gram(B) of another tensor ran in between?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd
from svdq_amd import workloads
from svdq_amd.pipeline import CompressPlan
dev = torch.device("cuda", 0)
N = 8
def mk(D, seed, ur=1024):
    bufs, views = workloads.synth_task_buffers([D], N, seed=seed, device=dev)
    plan = CompressPlan([D], N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev, unit_rows=ur)
    table = plan.pointer_table(views)
    plan.run(table); torch.cuda.synchronize()
    return plan, table, bufs
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
def timed(seq_before, target, reps=20):
    tot = 0.0
    for _ in range(reps):
        flush.add_(1.0)
        for pl in seq_before: pl[0].gram_center(pl[1])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); target[0].basis_project(target[1]); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
M = 1024 * 1024
for dA, dB in ((4 * M, 4 * M), (3 * M, 1 * M), (1 * M, 4 * M), (4 * M, 1 * M), (2 * M, 2 * M), (1 * M, 1 * M)):
    A = mk(dA, 1); B = mk(dB, 2); C = mk(dA, 3)
    adj = timed([A], A); mid = timed([A, B], A); cold = timed([C], A)
    print(f"A={dA*32/1e6:.0f}MB B={dB*32/1e6:.0f}MB  bp(A): adjacent {adj:.1f} us | gram(B) in between {mid:.1f} us | cold {cold:.1f} us", flush=True)
    del A, B, C
