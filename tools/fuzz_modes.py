"""Randomised cross-check of the schedules and input modes of svdq_compress: for random (N, sizes, fp16, centre, bits,
stages) the fused schedule, the gather mode (against compacted copies) and the minus-base mode (against ingest +
compress) must reproduce the four-launch path bit for bit.  Prints one line per case and a summary."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd as sq
from svdq_amd.pipeline import CompressPlan
from svdq_amd.mask_loader import MaskSet

dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rnd = random.Random(seed)
g = torch.Generator(device=dev).manual_seed(seed)


def same(a: CompressPlan, b: CompressPlan, counts=None):
    if not torch.equal(a.small, b.small):
        return "small buffers differ"
    sm = a.fetch_small()
    for p in range(a.P):
        rows = int(sm.rows[p])
        x = a.basis_tensors(p, int(sm.k[p]), int(sm.r[p]), rows)
        y = b.basis_tensors(p, int(sm.k[p]), int(sm.r[p]), rows)
        for u, v in zip(x, y):
            if (u is None) != (v is None) or (u is not None and not torch.equal(u, v)):
                return f"basis/mean of parameter {p} differs"
    return None


bad = 0
for c in range(cases):
    N = rnd.choice([1, 2, 3, 5, 8, 8, 8, 12, 16, 17, 20, 24, 32])
    P = rnd.randint(1, 5)
    sizes = [rnd.choice([1, 3, 7, 255, 256, 257, 1000, 4096, 5001, 65536 + rnd.randint(0, 9), rnd.randint(1, 300000)])
             for _ in range(P)]
    fp16, center = rnd.random() < 0.7, rnd.random() < 0.8
    bits, stages = rnd.choice([2, 4, 8]), rnd.choice([1, 2, 4])
    unit_rows = rnd.choice([0, 1024, 4096])
    kw = dict(energy_threshold=rnd.choice([0.5, 0.9, 0.99]), max_rank=rnd.choice([None, 2, 64]), center=center, fp16=fp16,
              low_bits=bits, rtvq_stages=stages, device=dev, unit_rows=unit_rows)
    base = [torch.randn(D, device=dev, generator=g) for D in sizes]
    lat = [torch.randn(D, 3, device=dev, generator=g) for D in sizes]
    deltas = [[0.01 * (lat[p] @ torch.randn(3, device=dev, generator=g)) + 0.002 * torch.randn(sizes[p], device=dev, generator=g)
               for _ in range(N)] for p in range(P)]
    ref = CompressPlan(sizes, N, **kw)
    ref.run(ref.pointer_table(deltas))
    msgs = []
    if N <= 16:
        fu = CompressPlan(sizes, N, flags=4 | (rnd.choice([0, 1, 3]) << 8), **kw)
        fu.run(fu.pointer_table(deltas))
        torch.cuda.synchronize()
        m = same(ref, fu)
        if m: msgs.append("fused: " + m)
    # minus-base: fine-tuned = base + delta is not exactly invertible in fp32, so compare with ingest of the same tensors
    fts = [[base[p] + deltas[p][t] for t in range(N)] for p in range(P)]
    eb = sq.ElementwiseBatch(sizes, N, dev)
    ing = eb.ingest(base, [f for fs in fts for f in fs])
    r2 = CompressPlan(sizes, N, **kw)
    r2.run(r2.pointer_table([ing[p * N:(p + 1) * N] for p in range(P)]))
    fb = CompressPlan(sizes, N, **kw)
    fb.run_from_base(fb.pointer_table(fts), torch.tensor([b.data_ptr() for b in base], dtype=torch.int64).to(dev))
    torch.cuda.synchronize()
    m = same(r2, fb)
    if m: msgs.append("from_base: " + m)
    eb.close()
    # gather vs compaction
    dens = rnd.choice([0.0, 0.05, 0.5, 0.94, 1.0])
    masks = [(torch.rand(D, device=dev, generator=g) < dens) for D in sizes]
    ms = MaskSet(sizes, dev)
    dt, _, ct, _ = ms.compact(masks, deltas, want_false=False)
    it, _, ct2, _ = ms.indices(masks, want_false=False)
    r3 = CompressPlan(sizes, N, **kw)
    r3.run(r3.pointer_table(dt), ct)
    ga = CompressPlan(sizes, N, **kw)
    ga.run_gather(ga.pointer_table(deltas), torch.tensor([x.data_ptr() for x in it], dtype=torch.int64).to(dev), ct2)
    torch.cuda.synchronize()
    m = same(r3, ga)
    if m: msgs.append("gather: " + m)
    status = "ok" if not msgs else "MISMATCH " + "; ".join(msgs)
    bad += bool(msgs)
    print(f"case {c:3d}: N={N:2d} sizes={sizes} fp16={fp16} center={center} bits={bits} stages={stages} dens={dens}: {status}",
          flush=True)
print(f"{cases - bad} / {cases} cases identical", flush=True)
sys.exit(1 if bad else 0)
