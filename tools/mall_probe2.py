"""A/B: is k_basis_project faster when k_gram has just streamed the SAME tensor (Infinity Cache reuse)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd
from svdq_amd import workloads
from svdq_amd.pipeline import CompressPlan
dev = torch.device("cuda", 0)
N = 8
def mk(D, seed, ur, flags=0):
    bufs, views = workloads.synth_task_buffers([D], N, seed=seed, device=dev)
    plan = CompressPlan([D], N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev, unit_rows=ur, flags=flags)
    table = plan.pointer_table(views)
    plan.run(table); torch.cuda.synchronize()
    return plan, table, bufs
for D in (1024 * 1024, 2048 * 1024, 3072 * 1024, 4096 * 1024, 6144 * 1024):
    ur = 1024
    A = mk(D, 1, ur); B = mk(D, 2, ur); R = mk(D, 1, ur, flags=1)
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
    def timed(first, second, reps=20):
        tot = 0.0
        for _ in range(reps):
            flush.add_(1.0)                       # evict everything
            first[0].gram_center(first[1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); second[0].basis_project(second[1]); e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps * 1e3
    same = timed(A, A); diff = timed(A, B); same2 = timed(B, B); diff2 = timed(B, A); rev = timed(R, R); rev2 = timed(R, R)
    print(f"D={D} in={D*N*4/1e6:.0f}MB out={D*(2*N+4)/1e6:.0f}MB  bp after gram(same) {same:.1f}/{same2:.1f} us   reversed {rev:.1f}/{rev2:.1f} us   after gram(other) {diff:.1f}/{diff2:.1f} us", flush=True)
    del A, B, R, flush
