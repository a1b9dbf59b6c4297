"""Wall time of the reference-shaped call (run_basis_and_compress: dictionaries in, dictionaries out) against the
GPU work inside it: how much is host-side assembly?  Also: everything touched (every task artifact of every
parameter materialised), and torch.ops.svdq.compress with its plan cache."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd as sq
from svdq_amd import workloads

dev = torch.device("cuda", 0)
model = sys.argv[1] if len(sys.argv) > 1 else "ViT-L-14"
N = 8
shapes = workloads.vit_visual_shapes(model)
names = sorted(shapes)
rows = [workloads.numel(shapes[n]) for n in names]
bufs, views = workloads.synth_task_buffers(rows, N, seed=1, device=dev)
tv = {f"task{t}": {n: views[p][t].view(shapes[n]) for p, n in enumerate(names)} for t in range(N)}
cfg = sq.SVDHybridConfig(svd_energy_threshold=0.9, svd_max_rank=64)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def touch_all():
    bases, comp = sq.run_basis_and_compress(tv, {}, cfg, "cuda")
    n = 0
    for name, per_task in comp.items():
        for t, art in per_task.items():
            n += len(art["masked"]["c_low_quant"]["payloads"])
    return n


print(f"{model} x {N}: run_basis_and_compress {timeit(lambda: sq.run_basis_and_compress(tv, {}, cfg, 'cuda')):.1f} ms per call",
      flush=True)
print(f"  ... and every task artifact of every parameter materialised: {timeit(touch_all):.1f} ms", flush=True)
from svdq_amd.pipeline import CompressPlan
pl = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, device=dev)
tab = pl.pointer_table(views)
print(f"pointer_table: {timeit(lambda: pl.pointer_table(views)):.2f} ms", flush=True)
print(f"run + fetch_small (GPU work + one D2H): {timeit(lambda: (pl.run(tab), pl.fetch_small())):.2f} ms", flush=True)
flat = [v for vs in views for v in vs]
print(f"torch.ops.svdq.compress (cached plan, no sync inside): "
      f"{timeit(lambda: torch.ops.svdq.compress(flat, N, 0.9, 64, True, True, 4, 2)):.2f} ms per call", flush=True)
pr = cProfile.Profile()
pr.enable()
bases, comp = sq.run_basis_and_compress(tv, {}, cfg, "cuda")
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
