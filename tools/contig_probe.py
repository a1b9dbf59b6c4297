"""Does a physically contiguous output allocation (hipExtMallocWithFlags, hipDeviceMallocContiguous) pin pass 2 to
its fast level?  Compares torch-allocated candidates with contiguous ones in one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svdq_amd
from svdq_amd import workloads
from svdq_amd.pipeline import CompressPlan

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]


class Raw:
    def __init__(self, nbytes, flags):
        self.ptr = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(self.ptr), nbytes, flags)
        if rc != 0:
            raise RuntimeError(f"hipExtMallocWithFlags({nbytes}, {flags}) -> {rc}")
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr.value, False), "version": 2}

    def free(self):
        hip.hipFree(self.ptr)


dev = torch.device("cuda", 0)
N = 8
shapes = workloads.vit_visual_shapes("ViT-L-14")
rows = [workloads.numel(shapes[k]) for k in sorted(shapes)]
bufs, views = workloads.synth_task_buffers(rows, N, seed=1, device=dev)
plan = CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, device=dev)
table = plan.pointer_table(views)
plan.run(table); torch.cuda.synchronize()


def timed(reps=10):
    plan.basis_project(table); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.basis_project(table)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


base_basis = plan.basis
nb = base_basis.numel()
print("torch-allocated original:", round(timed(), 4), flush=True)
keep = []
for i in range(6):
    t = torch.empty(nb + (i % 3) * (3 << 20), dtype=torch.uint8, device=dev)
    keep.append(t)
    plan.basis = t[:nb]
    print(f"torch candidate {i}: {timed():.4f}", flush=True)
for flags, name in ((0x4, "contiguous"), (0x0, "default flags")):
    for i in range(5):
        try:
            r = Raw(nb + (i % 3) * (3 << 20), flags)
        except RuntimeError as e:
            print(name, "failed:", e, flush=True)
            break
        keep.append(r)
        plan.basis = torch.as_tensor(r, device=dev)[:nb]
        print(f"{name} candidate {i} at {hex(r.ptr.value)}: {timed():.4f}", flush=True)
plan.basis = base_basis
print("original again:", round(timed(), 4), flush=True)
