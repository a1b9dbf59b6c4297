"""Steady-state pipelined order: ... G(k), B(k-1), G(k+1), B(k) ...: does B(k) still hit the Infinity Cache?"""
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
def timed(prep, target, reps=20):
    tot = 0.0
    for _ in range(reps):
        flush.add_(1.0)
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); target[0].basis_project(target[1]); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
M = 1024 * 1024
for d in (4 * M, 3 * M, 2 * M, 1 * M):
    A = mk(d, 1); B = mk(d, 2); C = mk(d, 3); E = mk(d, 4)
    g = lambda P: P[0].gram_center(P[1])
    b = lambda P: P[0].basis_project(P[1])
    r = {}
    r["adjacent G(A)"] = timed(lambda: g(A), A)
    r["G(A) G(B)"] = timed(lambda: (g(A), g(B)), A)
    r["G(C) G(A) B(C) G(B)  [d=1 steady state]"] = timed(lambda: (g(C), g(A), b(C), g(B)), A)
    r["G(A) B(C) G(B) B(E)... d=2-ish: G(A) G(B) B(C) G(E)"] = timed(lambda: (g(A), g(B), b(C), g(E)), A)
    r["cold"] = timed(lambda: g(C), A)
    print(f"tensor={d*32/1e6:.0f}MB  " + " | ".join(f"{k}: {v:.1f}" for k, v in r.items()), flush=True)
    del A, B, C, E
