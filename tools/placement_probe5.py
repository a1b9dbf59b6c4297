"""Copy time (svdq_hbm_probe mode 1, 1 GiB) against the DISTANCE between source and destination inside one huge
allocation: is the read/write interference level a function of address bits >= 30?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import c_void_p
import torch
from svdq_amd import _native as nat

dev = torch.device("cuda", 0)
lib = nat.lib()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
G = 1 << 30
H = torch.empty(160 * G, dtype=torch.uint8, device=dev)
H[:2 * G].fill_(1)
base = H.data_ptr()
print(f"one allocation of 160 GiB at VA {base:#x}", flush=True)


def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def copy(src_off, dst_off, nbytes=G):
    return ev(lambda: lib.svdq_hbm_probe(1, c_void_p(base + src_off), c_void_p(base + dst_off), nbytes, st))


for src in (0, 64 * G):
    line = []
    for d in range(0, 158):
        if abs(d * G - src) < G:
            line.append("  -  ")
            continue
        line.append(f"{copy(src, d * G):.3f}")
    print(f"src at +{src // G} GiB; dst at +d GiB, d = 0..157:", flush=True)
    for i in range(0, len(line), 16):
        print(f"  d={i:3d}: " + " ".join(line[i:i + 16]), flush=True)
# finer: distances around one transition, 64 MiB steps
M64 = 64 << 20
line = [f"{copy(0, 2 * G + i * M64):.3f}" for i in range(64)]
print("src +0; dst = +2 GiB + i * 64 MiB, i = 0..63:", flush=True)
for i in range(0, 64, 16):
    print("  " + " ".join(line[i:i + 16]), flush=True)
