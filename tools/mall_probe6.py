"""Upper bound of what the Infinity Cache could give a pipelined schedule: pass 1 and pass 2 of each ~128 MB group of
parameters launched back to back (pass 2 uses W of the previous step, which is the same W -- so there is NO eigen-stage
between them: zero-latency hand-over), against the same launches in the cache-hostile order (all pass-1 groups, then
all pass-2 groups) and against the two whole-model launches."""
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
st = torch.cuda.current_stream()


def ev(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for ur in (0, 2048, 1024):
    plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev, unit_rows=ur)
    table = plan.pointer_table(views)
    plan.tune_placement(table, candidates=6)
    plan.run(table); torch.cuda.synchronize()
    whole = ev(lambda: (plan.gram_center(table), plan.basis_project(table)))
    for mb in (64, 128, 256, 512):
        groups, p0, acc = [], 0, 0
        for p, d in enumerate(rows):
            acc += d * N * 4
            if acc >= mb << 20:
                groups.append((p0, p + 1 - p0)); p0, acc = p + 1, 0
        if p0 < len(rows):
            groups.append((p0, len(rows) - p0))

        def paired():
            for a, n in groups:
                plan.gram_range(table, a, n, st)
                plan.bp_range(table, a, n, st)

        def apart():
            for a, n in groups:
                plan.gram_range(table, a, n, st)
            for a, n in groups:
                plan.bp_range(table, a, n, st)

        print(f"unit_rows {ur or 4096}: groups of >= {mb} MB ({len(groups)} groups): pass 1 + pass 2 paired {ev(paired):.3f} ms, "
              f"same launches apart {ev(apart):.3f} ms, two whole-model launches {whole:.3f} ms", flush=True)
    plan.close()
