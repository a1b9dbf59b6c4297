"""Where a launch of the two streaming passes spends its time: ramp-up, steady state, tail.

Needs the diagnostic build of the library (csrc compiled with -DSVDQ_UNIT_STAMPS: every work unit records when its
wavefront started and ended and on which XCD): SVDQ_LIB_PATH=gpurun_ab/libsvdq_stamps.so python tools/unit_timeline.py
[--shard-of K] [--model M] [--tasks N] [--unit-rows R].  Prints, per pass: launch span, busy wave-time / (span x slots),
how many waves are resident over time (10 bins), start / end spreads, unit duration percentiles early vs late."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import svdq_amd as sq
from svdq_amd import workloads, shard

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="ViT-L-14")
ap.add_argument("--tasks", type=int, default=8)
ap.add_argument("--shard-of", type=int, default=8)
ap.add_argument("--unit-rows", type=int, default=0)
ap.add_argument("--flags", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib = sq._native.lib()
for f in ("svdq_debug_stamps_gram", "svdq_debug_stamps_project"):
    getattr(lib, f).argtypes = [ctypes.c_void_p]
    getattr(lib, f).restype = ctypes.c_int

shapes = workloads.vit_visual_shapes(args.model)
names = sorted(shapes)
rows_all = [workloads.numel(shapes[n]) for n in names]
mine = shard.partition_lpt(rows_all, args.shard_of)[0] if args.shard_of > 1 else list(range(len(names)))
rows = [rows_all[i] for i in mine]
N = args.tasks
bufs, views = workloads.synth_task_buffers(rows, N, seed=1234, device=dev)
plan = sq.CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, low_bits=4, rtvq_stages=2,
                       device=dev, unit_rows=args.unit_rows, flags=args.flags)
table = plan.pointer_table(views)
nu = int(plan.sizes.n_units)
print(json.dumps({"tensors": len(rows), "sum_rows": int(sum(rows)), "units": nu, "tasks": N}))
sg = torch.zeros(3 * nu, dtype=torch.int64, device=dev)
sp = torch.zeros(3 * nu, dtype=torch.int64, device=dev)
for _ in range(5):
    plan.run(table)
torch.cuda.synchronize()
assert lib.svdq_debug_stamps_gram(sg.data_ptr()) == 0 and lib.svdq_debug_stamps_project(sp.data_ptr()) == 0
for _ in range(3):          # the step that is looked at is the last of three back to back, like a timed step
    plan.run(table)
torch.cuda.synchronize()
lib.svdq_debug_stamps_gram(None)
lib.svdq_debug_stamps_project(None)


def report(name, st, slots_per_cu):
    s = st.cpu().numpy().reshape(nu, 3)
    t0, t1, xcc = s[:, 0].astype(np.float64), s[:, 1].astype(np.float64), s[:, 2] & 0xf
    ok = t1 > 0
    t0, t1, xcc = t0[ok], t1[ok], xcc[ok]
    base = t0.min()
    a, b = (t0 - base) / 100.0, (t1 - base) / 100.0        # microseconds (100 MHz ticks)
    span = b.max()
    dur = b - a
    slots = 256 * slots_per_cu
    print(f"== {name}: {len(a)} units, span {span:.1f} us, wave-time / (span x {slots} slots) = {dur.sum() / (span * slots):.3f}")
    print(f"   starts: 50 % by {np.percentile(a, 50):.1f} us, 90 % by {np.percentile(a, 90):.1f}, last {a.max():.1f};"
          f"  ends: first {b.min():.1f}, 10 % by {np.percentile(b, 10):.1f}, 50 % {np.percentile(b, 50):.1f}, 90 % {np.percentile(b, 90):.1f}")
    edges = np.linspace(0, span, 21)
    res = [int(((a <= 0.5 * (edges[i] + edges[i + 1])) & (b > 0.5 * (edges[i] + edges[i + 1]))).sum()) for i in range(20)]
    print("   resident waves at the middle of 20 equal time bins:", res)
    first = a < np.percentile(a, 25)
    last = a > np.percentile(a, 75)
    full = dur > 0.5 * np.median(dur)
    for lab, m in (("first quarter of the starts", first & full), ("last quarter", last & full), ("all", full)):
        d = dur[m]
        if len(d):
            print(f"   unit duration, {lab}: median {np.median(d):.1f} us, p10 {np.percentile(d, 10):.1f}, p90 {np.percentile(d, 90):.1f}, max {d.max():.1f}")
    per = [int((xcc == x).sum()) for x in range(8)]
    endx = [round(float(b[xcc == x].max()), 1) if per[x] else None for x in range(8)]
    print(f"   units per XCD {per}; last end per XCD {endx}")


report("pass 1 (k_gram)", sg, 5 * 4)
report("pass 2 (k_basis_project)", sp, 5 * 4)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    plan.run(table)
e1.record()
torch.cuda.synchronize()
print(f"step (stamps off): {e0.elapsed_time(e1) / 50:.4f} ms")
