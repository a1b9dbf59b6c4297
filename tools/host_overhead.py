"""Wall time of the reference-shaped call (run_basis_and_compress: dictionaries in, dictionaries out) against the
5 ms of GPU work inside it: how much is host-side assembly?"""
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
for _ in range(2):
    bases, comp = sq.run_basis_and_compress(tv, {}, cfg, "cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    bases, comp = sq.run_basis_and_compress(tv, {}, cfg, "cuda")
torch.cuda.synchronize()
print(f"{model}: run_basis_and_compress {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per call", flush=True)
from svdq_amd.pipeline import CompressPlan
t0 = time.perf_counter()
for _ in range(5):
    pl = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, device=dev)
    pl.close()
    del pl
torch.cuda.synchronize()
print(f"CompressPlan create + close: {1e3 * (time.perf_counter() - t0) / 5:.1f} ms", flush=True)
pl = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, device=dev)
tab = pl.pointer_table(views)
t0 = time.perf_counter()
for _ in range(5):
    tab = pl.pointer_table(views)
print(f"pointer_table: {1e3 * (time.perf_counter() - t0) / 5:.1f} ms", flush=True)
t0 = time.perf_counter()
for _ in range(5):
    pl.run(tab)
    sm = pl.fetch_small()
print(f"run + fetch_small: {1e3 * (time.perf_counter() - t0) / 5:.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
bases, comp = sq.run_basis_and_compress(tv, {}, cfg, "cuda")
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
