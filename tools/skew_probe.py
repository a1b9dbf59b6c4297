"""Does the relative placement of the N task buffers matter for pass 1?  The N streams are read in lockstep (same
parameter, same rows, N base addresses); this times k_gram with task buffer t shifted by t * skew bytes inside its own
allocation.  python tools/skew_probe.py [model] [tasks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import svdq_amd
from svdq_amd import workloads
from svdq_amd.pipeline import CompressPlan

dev = torch.device("cuda", 0)
model = sys.argv[1] if len(sys.argv) > 1 else "ViT-L-14"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shapes = workloads.vit_visual_shapes(model)
rows = [workloads.numel(shapes[n]) for n in sorted(shapes)]
offs, tot = [], 0
for d in rows:
    offs.append(tot)
    tot += (d + 63) // 64 * 64
plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, low_bits=4, rtvq_stages=2, device=dev)
g = torch.Generator(device=dev).manual_seed(3)
pad = (8 << 20) // 4 * N          # room for the largest skew
raw = [torch.empty(tot + pad, device=dev).normal_(generator=g) for _ in range(N)]
print("base addresses mod 2 MiB / 1 GiB:", [(b.data_ptr() % (2 << 20), (b.data_ptr() >> 30)) for b in raw])


def ev_time(fn, reps=8):
    for _ in range(2):
        fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(e[i].elapsed_time(e[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


for skew in (0, 256, 0, 64, 0, 4096, 0, 128, 512, 0):
    sh = [t * skew // 4 for t in range(N)]
    views = [[raw[t][sh[t] + o:sh[t] + o + d] for t in range(N)] for d, o in zip(rows, offs)]
    table = plan.pointer_table(views)
    t1 = ev_time(lambda: plan.gram_center(table))
    plan.gram_center(table)
    plan.eig_rank_select(table)
    t2 = ev_time(lambda: plan.basis_project(table))
    print(f"skew {skew:>9d} B per task: pass 1 {t1:.4f} ms   pass 2 {t2:.4f} ms", flush=True)
