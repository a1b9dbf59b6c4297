"""What the per-step exchange of the small artifacts costs one rank of the 8-GPU run (one-rank RCCL world on one GPU):
A = the compress step alone, B = + the copy into the send slot, C = + the planned async all-gather (RaggedGather, what
bench.py does), D = all-gather straight from the plan's buffer (no copy; not safe without a second small buffer),
E = C with the collective only every step's host call skipped (copy only + enqueue cost estimate)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import svdq_amd as sq
from svdq_amd import workloads, shard

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = workloads.vit_visual_shapes("ViT-L-14")
names = sorted(shapes)
rows_all = [workloads.numel(shapes[n]) for n in names]
rows = [rows_all[i] for i in shard.partition_lpt(rows_all, K)[0]]
N = 8
bufs, views = workloads.synth_task_buffers(rows, N, seed=1234, device=dev)
plan = sq.CompressPlan(rows, N, energy_threshold=0.9, max_rank=64, center=True, fp16=True, low_bits=4, rtvq_stages=2, device=dev)
table = plan.pointer_table(views)
rg = shard.RaggedGather(plan.small.numel(), dev)
print(f"shard of {K}: {len(rows)} tensors, small buffer {plan.small.numel()} bytes")


def timed(fn, steps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    t_host = (time.perf_counter() - t0) / steps * 1e3
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, t_host


def A():
    plan.run(table)


def B():
    plan.run(table)
    rg.send[:rg.nbytes].copy_(plan.small.view(torch.uint8).reshape(-1), non_blocking=True)


def C():
    plan.run(table)
    rg.run(plan.small, overlap=True)


def F():
    """C, but a set whose previous collective the host already sees completed is reused without a stream-level wait"""
    plan.run(table)
    if len(rg._sets) == 1:
        rg.run(plan.small, overlap=True)
        return
    nxt = (rg._cur + 1) % 2
    w = rg._work[nxt]
    if w is not None and w.is_completed():
        rg._work[nxt] = None
    rg.run(plan.small, overlap=True)


side = torch.cuda.Stream(device=dev)
evs = [torch.cuda.Event() for _ in range(4)]
tiny_src = torch.zeros(rg.stride, dtype=torch.uint8, device=dev)
tiny_dst = torch.zeros(rg.stride, dtype=torch.uint8, device=dev)
ei = [0]


def E1():
    """the step + an event recorded on its stream (nothing waits for it)"""
    plan.run(table)
    ei[0] = (ei[0] + 1) % 4
    evs[ei[0]].record()


def E2():
    """the step + event + a 10-KB copy kernel on a side stream that waits for the event (a stand-in for the collective)"""
    plan.run(table)
    ei[0] = (ei[0] + 1) % 4
    evs[ei[0]].record()
    side.wait_event(evs[ei[0]])
    with torch.cuda.stream(side):
        tiny_dst.copy_(tiny_src, non_blocking=True)


def S():
    """the step + a BLOCKING all-gather (the stream waits for it)"""
    plan.run(table)
    rg.run(plan.small, overlap=False)


recv2 = [torch.empty_like(rg.recv), torch.empty_like(rg.recv)]
pad = torch.zeros(rg.stride, dtype=torch.uint8, device=dev)
works = [None, None]
cur = [0]


def D():
    plan.run(table)
    cur[0] ^= 1
    if works[cur[0]] is not None:
        works[cur[0]].wait()
    src = plan.small.view(torch.uint8).reshape(-1)
    works[cur[0]] = dist.all_gather_into_tensor(recv2[cur[0]][:src.numel()], src, async_op=True)


for name, fn in (("A step alone", A), ("B + copy into the send slot", B), ("C + planned async all-gather (bench.py)", C),
                 ("D all-gather straight from plan.small", D), ("F = C without the stream wait when already complete", F),
                 ("E1 step + event record", E1), ("E2 step + event + side-stream copy", E2), ("S step + blocking all-gather", S),
                 ("A again", A), ("C again", C)):
    ms, host = timed(fn)
    rg.finish()
    for w in works:
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    print(f"{name:48s} {ms:.4f} ms per step on the GPU, host enqueue {host:.4f} ms per step")
dist.destroy_process_group()
