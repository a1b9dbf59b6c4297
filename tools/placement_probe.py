"""Does the relative placement of the output buffers (basis, mean) against the task buffers change pass 2's
speed?  bench.py shows k_basis_project at either ~2.75 or ~3.0 ms from one process to the next."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd
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
print("task buffers at", [hex(b.data_ptr()) for b in bufs], flush=True)
print("basis", hex(plan.basis.data_ptr()), "mean", hex(plan.mean.data_ptr()), "ws", hex(plan.workspace.data_ptr()), flush=True)

def timed(reps=10):
    plan.basis_project(table); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.basis_project(table)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

base_basis, base_mean = plan.basis, plan.mean
print("as allocated:", round(timed(), 4), flush=True)
MB = 1 << 20
def tz(x):
    return (x & -x).bit_length() - 1
print("basis align 2^%d, mean 2^%d" % (tz(base_basis.data_ptr()), tz(base_mean.data_ptr())), flush=True)
keep = []
def fill_ms(buf, reps=3):
    buf.fill_(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        buf.fill_(1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def read_ms(buf, reps=3):
    v = buf.view(torch.int32)
    v.sum(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        v.sum()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def bp_variants():
    full = timed()
    m = plan.mean
    plan.mean = None            # NULL mean pointer: pass 2 skips the mean stream
    nomean = timed()
    plan.mean = m
    return full, nomean
print("original basis: fill %.4f ms, bp %.4f / without the mean stream %.4f ms" % ((fill_ms(base_basis),) + bp_variants()), flush=True)
for i in range(10):
    extra = (i % 4) * 3
    big = torch.empty(base_basis.numel() + extra * MB, dtype=torch.uint8, device=dev)
    keep.append(big)
    plan.basis = big[:base_basis.numel()]
    full, nomean = bp_variants()
    print(f"candidate {i:2d} at {hex(big.data_ptr())}: bp {full:.4f} ms   without the mean stream {nomean:.4f} ms", flush=True)
plan.basis = base_basis
mean0 = plan.mean
for i in range(6):
    mm = torch.empty(mean0.numel() + (i % 3) * MB, dtype=torch.float32, device=dev)
    keep.append(mm)
    plan.mean = mm[:mean0.numel()]
    print(f"mean candidate {i} at {hex(mm.data_ptr())}: bp {timed():.4f} ms", flush=True)
plan.mean = mean0
# one allocation holding both outputs: basis first, mean behind it at a few different distances
nb = base_basis.numel()
nm = mean0.numel() * 4
for i in range(10):
    pad = (0, 256, 4096, 1 << 20, 3 << 20)[i % 5]
    off = (nb + 255) // 256 * 256 + pad
    both = torch.empty(off + nm + (i // 5) * 5 * MB, dtype=torch.uint8, device=dev)
    keep.append(both)
    plan.basis = both[:nb]
    plan.mean = both[off:off + nm].view(torch.float32)
    print(f"joint candidate {i:2d} at {hex(both.data_ptr())} pad {pad:>8}: bp {timed():.4f} ms", flush=True)
plan.basis = base_basis
plan.mean = mean0
print("original again: bp %.4f" % timed(), flush=True)
