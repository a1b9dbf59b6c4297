"""Is the 'reuse' benefit data residency (Infinity Cache) or translation warmth (TLB)?"""
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
def timed(prep, target, reps=20, what="bp"):
    tot = 0.0
    for _ in range(reps):
        flush.add_(1.0)
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if what == "bp": target[0].basis_project(target[1])
        else: target[0].gram_center(target[1])
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
M = 1024 * 1024
sink = torch.zeros(1, device=dev)
for dA in (4 * M, 2 * M):
    A = mk(dA, 1); C = mk(dA, 3)
    def touch_in(stride):
        for b in A[2]: sink.add_(b[::stride].sum())
    def touch_out(stride):
        sink.add_(A[0].basis[::stride * 4].sum().float()); sink.add_(A[0].mean[::stride].sum())
    r = {}
    r["cold"] = timed(lambda: C[0].gram_center(C[1]), A)
    r["gram(A) first"] = timed(lambda: A[0].gram_center(A[1]), A)
    r["touch in 1/4KB"] = timed(lambda: touch_in(1024), A)
    r["touch in+out 1/4KB"] = timed(lambda: (touch_in(1024), touch_out(1024)), A)
    r["touch in 1/64KB"] = timed(lambda: touch_in(16384), A)
    r["touch in 1/2MB"] = timed(lambda: touch_in(524288), A)
    r["gram cold"] = timed(lambda: C[0].gram_center(C[1]), A, what="gram")
    r["gram after touch in 1/4KB"] = timed(lambda: touch_in(1024), A, what="gram")
    r["gram after gram(A)"] = timed(lambda: A[0].gram_center(A[1]), A, what="gram")
    print(f"A={dA*32/1e6:.0f}MB  " + " | ".join(f"{k}: {v:.1f}" for k, v in r.items()), flush=True)
    del A, C
