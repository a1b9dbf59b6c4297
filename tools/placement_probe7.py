"""Would U_high and U_low in DIFFERENT memory regions help pass 2 further?  Experiment build only
(svd-quantization-task-merging_amd/var_ulow.so: svdq_project.hip compiled with -DSVDQ_EXP_ULOW_SHIFT=8 GiB, so U_low of every
parameter is written 8 GiB behind its slab), selected with SVDQ_LIB_PATH.  Same harness as placement_probe6.py: one
huge allocation cut into 4 GiB slots, slots classified by the copy probe; the basis then starts in a slot o whose
slots o, o+1 lie in one region and o+2, o+3 in another, the mean and the deltas go into chosen regions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import c_void_p
import torch
from svdq_amd import workloads, _native as nat
from svdq_amd.pipeline import CompressPlan

assert "var_ulow" in nat.LIB_PATH, "run with SVDQ_LIB_PATH=.../var_ulow.so"
dev = torch.device("cuda", 0)
lib = nat.lib()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
G = 1 << 30
SLOT, NSLOT = 4 * G, 60
H = torch.empty(NSLOT * SLOT, dtype=torch.uint8, device=dev)
base = H.data_ptr()
L = "ABCDEFGH"


def ev(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def copy_ms(a, b):
    return ev(lambda: lib.svdq_hbm_probe(1, c_void_p(base + a * SLOT), c_void_p(base + b * SLOT + 2 * G), G, st))


ts = sorted(copy_ms(0, x) for x in range(1, NSLOT, 3))
thr = 0.5 * (ts[0] + ts[-1])
refs, label = [], [-1] * NSLOT
for s in range(NSLOT):
    for ri, r in enumerate(refs):
        if r == s or copy_ms(r, s) > thr:
            label[s] = ri
            break
    else:
        refs.append(s); label[s] = len(refs) - 1
print("regions of the 4 GiB slots:", "".join(L[l] for l in label), flush=True)

N = 8
shapes = workloads.vit_visual_shapes("ViT-L-14")
rows = [workloads.numel(shapes[k]) for k in sorted(shapes)]
offs, tot = [], 0
for d in rows:
    offs.append(tot); tot += (d + 63) // 64 * 64
Hf = H.view(torch.float32)
src_bufs, _ = workloads.synth_task_buffers(rows, N, seed=1, device=dev)
plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev)
bb = plan.sizes.basis_bytes
assert bb <= 2 * SLOT


def find_basis_slot(high, low, used):
    for o in range(NSLOT - 3):
        if label[o] == label[o + 1] == high and label[o + 2] == label[o + 3] == low and not (set(range(o, o + 4)) & used):
            return o
    return None


def run(in_label, high, low, mean_label):
    used = set()
    o = find_basis_slot(high, low, used)
    if o is None:
        print(f"no slot run for U_high {L[high]} / U_low {L[low]}"); return
    used |= set(range(o, o + 4))
    m = next((x for x in range(NSLOT) if label[x] == mean_label and x not in used), None)
    used.add(m)
    ins = []
    for x in range(NSLOT):
        if label[x] == in_label and x not in used and len(ins) < N:
            ins.append(x); used.add(x)
    if m is None or len(ins) < N:
        print("not enough slots"); return
    bufs = []
    for t in range(N):
        b = Hf[ins[t] * SLOT // 4: ins[t] * SLOT // 4 + tot]
        b.copy_(src_bufs[t]); bufs.append(b)
    views = [[bufs[t][oo:oo + d] for t in range(N)] for d, oo in zip(rows, offs)]
    plan.basis = H[o * SLOT:o * SLOT + bb]              # U_low lands 8 GiB further: slots o+2, o+3
    plan.mean = Hf[m * SLOT // 4: m * SLOT // 4 + plan.sizes.mean_floats]
    plan._typed = None
    table = plan.pointer_table(views)
    plan.run(table); torch.cuda.synchronize()
    t2 = ev(lambda: plan.basis_project(table), 6)
    print(f"deltas {L[in_label]}  U_high {L[high]}  U_low {L[low]}  mean {L[mean_label]}:  pass 2 {t2:.3f} ms", flush=True)


labs = sorted(set(label), key=lambda l: -label.count(l))[:3]
a, b, c = labs
run(a, b, b, b)      # everything written into one region (reference: 2.93)
run(a, b, b, c)      # basis together, mean apart (2.74)
run(a, b, c, c)      # U_high apart from U_low, mean with U_low
run(a, b, c, b)      # U_high + mean together, U_low apart
run(a, b, c, a)      # three write streams: U_high B, U_low C, mean with the deltas
run(a, c, b, a)
run(a, a, b, c)      # U_high with the deltas
run(a, b, a, c)      # U_low with the deltas
