"""Bandwidth of the kernels around the headline path (SURVEY.md 8(d) 'standalone large-tensor form', 8 f1/f4):
standalone RTVQ, batched whole-tensor quantization, task-vector ingest, task Gram, reconstruct.
Prints one JSON object per line; algorithmic bytes are stated per entry."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd as sq
from svdq_amd import workloads
from svdq_amd.rtvq import RTVQQuantizer

dev = torch.device("cuda", 0)


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def emit(name, ms, alg_bytes, practical_bytes=None, **kw):
    d = {"kernel": name, "ms": round(ms, 4), "algorithmic_GBps": round(alg_bytes / ms / 1e6, 1)}
    if practical_bytes:
        d["moved_GBps"] = round(practical_bytes / ms / 1e6, 1)
    d.update(kw)
    print(json.dumps(d), flush=True)


# ---- standalone multi-stage quantizer (K3): n*(4+S) algorithmic, n*(4(S+1)+S)-ish moved
lib = sq._native.lib()
from svdq_amd.pipeline import _ptr, _stream_ptr
for n in (589_824, 4_194_304, 67_108_864):
    x = 0.01 * torch.randn(n, device=dev)
    for bits, S in ((4, 2), (4, 4)):
        work = torch.empty(int(lib.svdq_rtvq_work_bytes(n)), dtype=torch.uint8, device=dev)
        codes = torch.empty((S, n), dtype=torch.uint8, device=dev)
        sc, zp, rn = (torch.empty(S, dtype=torch.float32, device=dev) for _ in range(3))
        fn = lambda: lib.svdq_rtvq_quantize(_ptr(x), n, bits, S, _ptr(codes), n, _ptr(sc), _ptr(zp), _ptr(rn), _ptr(work),
                                            _stream_ptr())
        ms = timed(fn)
        # moved: stats read 4n; stage s: read 4n, write n codes (+ 4n residual unless last)
        moved = 4 * n + sum(4 * n + n + (4 * n if s < S - 1 else 0) for s in range(S))
        emit("svdq_rtvq_quantize", ms, n * (4 + S), moved, n=n, bits=bits, stages=S, launches=1 + S)

# ---- batched ingest / TVQ / task Gram on ViT-L-14 x 8
shapes = workloads.vit_visual_shapes("ViT-L-14")
names = sorted(shapes)
rows = [workloads.numel(shapes[k]) for k in names]
N = 8
total = sum(rows)
batch = sq.ElementwiseBatch(rows, N, dev)
base = [torch.randn(r, device=dev) for r in rows]
ft = [base[p] + 0.01 * torch.randn(rows[p], device=dev) for p in range(len(rows)) for _ in range(N)]
deltas = batch.ingest(base, ft)
tb = torch.tensor([t.data_ptr() for t in base], dtype=torch.int64).to(dev)
tf = torch.tensor([t.data_ptr() for t in ft], dtype=torch.int64).to(dev)
td = torch.tensor([t.data_ptr() for t in deltas], dtype=torch.int64).to(dev)
h = batch.plan._h
from ctypes import c_void_p
ms = timed(lambda: lib.svdq_ingest(h, _ptr(tb), _ptr(tf), _ptr(td), c_void_p(0), _stream_ptr()))
emit("svdq_ingest", ms, total * 4 * (2 * N + 1), tensors=len(rows), tasks=N, note="read base once + N finetuned, write N deltas")
ms = timed(lambda: lib.svdq_ingest(h, _ptr(tb), _ptr(tf), _ptr(td), _ptr(batch.work), _stream_ptr()))
emit("svdq_ingest+stats", ms, total * 4 * (2 * N + 1), tensors=len(rows), tasks=N)
del ft
codes = [torch.empty(rows[i // N], dtype=torch.uint8, device=dev) for i in range(len(rows) * N)]
tc = torch.tensor([t.data_ptr() for t in codes], dtype=torch.int64).to(dev)
sc = torch.empty(len(rows) * N, dtype=torch.float32, device=dev)
zp = torch.empty_like(sc)
for mode, nm in ((0, "asymmetric"), (1, "absmax")):
    ms = timed(lambda: lib.svdq_tvq_quantize(h, _ptr(td), mode, 8, _ptr(tc), _ptr(sc), _ptr(zp), _ptr(batch.work), 0,
                                             _stream_ptr()))
    emit(f"svdq_tvq_quantize[{nm},8b]", ms, total * N * 5, total * N * 9, tensors=len(rows) * N, launches=3,
         note="algorithmic: read 4 B + write 1 B per scalar; moved: + statistics pass")
    ms = timed(lambda: lib.svdq_tvq_quantize(h, _ptr(td), mode, 8, _ptr(tc), _ptr(sc), _ptr(zp), _ptr(batch.work), 1,
                                             _stream_ptr()))
    emit(f"svdq_tvq_quantize[{nm},8b,stats from ingest]", ms, total * N * 5, total * N * 5, launches=2)
ms = timed(lambda: lib.svdq_tvq_dequantize(h, _ptr(tc), 0, _ptr(sc), _ptr(zp), c_void_p(0), _ptr(td), _stream_ptr()))
emit("svdq_tvq_dequantize", ms, total * N * 5)
plan = sq.CompressPlan(rows, N, center=False, device=dev, gram_only=True)
table = td
G = torch.empty((N, N), dtype=torch.float64, device=dev)
ms = timed(lambda: lib.svdq_task_gram(plan._h, _ptr(table), c_void_p(0), _ptr(plan.workspace), _ptr(G), _stream_ptr()))
emit("svdq_task_gram", ms, total * N * 4, tensors=len(rows), tasks=N, launches=3)

# ---- the same at 20 tasks (BASELINE config #5: cluster weighting needs the Gram of the concatenated task vectors)
del codes, deltas, base, batch, plan
torch.cuda.empty_cache()
N20 = 20
bufs20, views20 = workloads.synth_task_buffers(rows, N20, seed=3, device=dev)
plan20 = sq.CompressPlan(rows, N20, center=False, device=dev, gram_only=True)
t20 = plan20.pointer_table(views20)
G20 = torch.empty((N20, N20), dtype=torch.float64, device=dev)
ms = timed(lambda: lib.svdq_task_gram(plan20._h, _ptr(t20), c_void_p(0), _ptr(plan20.workspace), _ptr(G20), _stream_ptr()))
emit("svdq_task_gram", ms, total * N20 * 4, tensors=len(rows), tasks=N20, launches=3)
import time
from svdq_amd import clustering
names20 = [f"t{i:02d}" for i in range(N20)]
Gh = G20.cpu().numpy()
for label in ("first call (imports sklearn)", "warm"):
    t0 = time.perf_counter()
    lab = clustering.cluster_from_gram(Gh, names20, 2, "kmeans")
    print(json.dumps({"host": f"clustering.cluster_from_gram(kmeans, k=2) on the 20 x 20 Gram, {label}",
                      "ms": round(1e3 * (time.perf_counter() - t0), 2)}), flush=True)
