"""Do the placement levels of pass 2 show up with plain streaming kernels too?  For each candidate output allocation:
pass 2, a device copy inputs -> candidate (svdq_hbm_probe mode 1, four 1.2 GB launches), the 8:5 mix probe (mode 2),
a torch fill of the candidate, and a read of the candidate (mode 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import c_void_p
import torch
from svdq_amd import workloads, _native as nat
from svdq_amd.pipeline import CompressPlan

dev = torch.device("cuda", 0)
N = 8
shapes = workloads.vit_visual_shapes("ViT-L-14")
rows = [workloads.numel(shapes[k]) for k in sorted(shapes)]
bufs, views = workloads.synth_task_buffers(rows, N, seed=1, device=dev)
plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev)
table = plan.pointer_table(views)
plan.run(table); torch.cuda.synchronize()
lib = nat.lib()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
nb = bufs[0].numel() * 4 // 16 * 16
sink = torch.empty(1 << 20, dtype=torch.float32, device=dev)


def ev(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def probe(mode, cand):
    def run():
        for t in range(4):
            if mode == 0:
                lib.svdq_hbm_probe(0, c_void_p(cand.data_ptr() + t * nb), c_void_p(sink.data_ptr()), nb, st)
            else:
                off = t * (nb if mode == 1 else nb // 8 * 5)
                lib.svdq_hbm_probe(mode, c_void_p(bufs[t].data_ptr()), c_void_p(cand.data_ptr() + off), nb, st)
    return run


keep = []
print("cand   pass2    copy(4x1.2GB)  mix8:5   fill    read-back   [ms]", flush=True)
for c in range(16):
    b, m = plan._alloc_outputs()
    keep.append((b, m))
    plan.basis, plan.mean, plan._typed = b, m, None
    t2 = ev(lambda: plan.basis_project(table), 6)
    tc = ev(probe(1, b))
    tm = ev(probe(2, b))
    tf = ev(lambda: b.fill_(0))
    tr = ev(probe(0, b))
    print(f"{c:3d}   {t2:.3f}    {tc:.3f}         {tm:.3f}    {tf:.3f}   {tr:.3f}", flush=True)
