"""
N > 1 path on CPU: two processes over gloo (127.0.0.1).  Parameter tensors are independent units, so
the only communication is the all-gather of the packed small-artifact buffers (ragged lengths).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _same_partition(a, b):
    return len({(int(x), int(y)) for x, y in zip(a, b)}) == len(set(int(x) for x in a)) == len(set(int(y) for y in b))


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import svdq_amd
    from svdq_amd import shard, workloads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = workloads.vit_visual_shapes("ViT-B-32")
        names = sorted(shapes)
        rows = [workloads.numel(shapes[n]) for n in names]
        parts = shard.partition_lpt(rows, world)
        mine = parts[rank]
        # every rank computes the same partition; shards are disjoint and cover everything
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(rows)))
        # ragged "small artifact" payload: 3 bytes per owned tensor, value = tensor index
        payload = torch.tensor([i % 251 for i in mine for _ in range(3)], dtype=torch.uint8)
        got = shard.gather_small(payload)
        assert len(got) == world
        for r, buf in enumerate(got):
            want = torch.tensor([i % 251 for i in parts[r] for _ in range(3)], dtype=torch.uint8)
            assert torch.equal(buf, want), r
        # the planned form bench.py uses inside its timed region: sizes exchanged once in the constructor, then one
        # all_gather_into_tensor per step into the same preallocated buffer -- new contents every step
        rg = shard.RaggedGather(payload.numel(), "cpu")
        assert rg.sizes == [3 * len(p) for p in parts]
        recv_ptr = rg.recv.data_ptr()
        for step in range(3):
            rg.run((payload + step).to(torch.uint8))
            assert rg.recv.data_ptr() == recv_ptr                    # nothing reallocated between steps
            for r, buf in enumerate(rg.views()):
                want = torch.tensor([(i % 251 + step) % 256 for i in parts[r] for _ in range(3)], dtype=torch.uint8)
                assert torch.equal(buf, want), (step, r)
        # the pipelined form: asynchronous collectives on two alternating buffer sets, waited for on reuse / finish()
        for step in range(5):
            rg.run((payload + 10 + step).to(torch.uint8), overlap=True)
        rg.finish()
        for r, buf in enumerate(rg.views()):
            want = torch.tensor([(i % 251 + 14) % 256 for i in parts[r] for _ in range(3)], dtype=torch.uint8)
            assert torch.equal(buf, want), ("overlap", r)
        try:
            rg.run(payload[:-1])
            raise AssertionError("a buffer of another length must be refused")
        except ValueError:
            pass
        loads = [sum(rows[i] for i in p) for p in parts]
        tot = torch.tensor([float(sum(rows[i] for i in mine))])
        dist.all_reduce(tot)
        assert int(tot.item()) == sum(rows)
        # cluster weighting: per-rank Gram of the owned parameters, one N*N all-reduce, same labels everywhere
        from helpers import load_golden
        from svdq_amd import clustering
        g = load_golden("cluster.npz")
        tasks = [str(t) for t in g["tasks"]]
        pnames = [str(p) for p in g["params"]]
        cparts = shard.partition_lpt([g[f"in__{p}"].shape[1] for p in pnames], world)
        G = torch.zeros((len(tasks), len(tasks)), dtype=torch.float64)
        for i in cparts[rank]:
            X = g[f"in__{pnames[i]}"].astype(np.float64).copy()
            if pnames[i] == str(g["missing"][1]):
                X[tasks.index(str(g["missing"][0]))] = 0.0
            G += torch.from_numpy(X @ X.T)
        shard.all_reduce_gram(G)
        for method in ("kmeans", "hierarchical"):
            for k in (2, 3):
                lab = clustering.cluster_from_gram(G.numpy(), tasks, k, method)
                assert _same_partition([lab[t] for t in tasks], g[f"labels__{method}__k{k}"]), (method, k)
        q.put((rank, "ok", loads))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, f"fail: {e!r}", None))
    finally:
        dist.destroy_process_group()


def test_two_rank_partition_and_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status, loads in res:
        assert status == "ok", (rank, status)
    loads = res[0][2]
    assert max(loads) / min(loads) < 1.05      # LPT keeps the two shards within 5 %


def test_partition_properties():
    import sys
    sys.path.insert(0, ROOT)
    from svdq_amd import shard, workloads
    for model, world in (("ViT-L-14", 8), ("ViT-B-16", 4), ("ViT-B-32", 3)):
        shapes = workloads.vit_visual_shapes(model)
        rows = [workloads.numel(shapes[n]) for n in sorted(shapes)]
        parts = shard.partition_lpt(rows, world)
        assert sorted(i for p in parts for i in p) == list(range(len(rows)))
        loads = [sum(rows[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(rows)          # never worse than one tensor apart
        assert parts == shard.partition_lpt(rows, world)     # deterministic
    assert shard.partition_lpt([5, 3], 4) == [[0], [1], [], []]
    # SURVEY section 8 shape totals
    tot = {m: sum(workloads.numel(s) for s in workloads.vit_visual_shapes(m).values()) for m in workloads.VIT_SPECS}
    assert tot == {"ViT-B-32": 87849216, "ViT-B-16": 86192640, "ViT-L-14": 303966208}
    assert {m: len(workloads.vit_visual_shapes(m)) for m in workloads.VIT_SPECS} == \
        {"ViT-B-32": 152, "ViT-B-16": 152, "ViT-L-14": 296}


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` without a launcher spawns its ranks itself; with fewer than N GPUs visible (none
    here) it must say so and exit non-zero BEFORE touching a GPU instead of silently running one rank."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SVDQ_DIST_BACKEND"):
        env.pop(k, None)
    if torch.cuda.device_count() >= 4:
        pytest.skip("needs a machine with fewer than 4 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "needs 4 GPUs" in r.stderr


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_takes_the_rccl_code_path():
    """`SVDQ_DIST_BACKEND=gloo python bench.py --gpus 2` on one card: the same Workload.step as the RCCL run -- LPT shard per
    rank, `RaggedGather.run(plan.small, overlap=True)` on the DEVICE buffer (the gather stages through pinned host memory
    by itself for this backend), barrier + max-over-ranks timing -- and the line carries every rank's own stage times and
    row share, which must be partition_lpt's."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    from svdq_amd import shard, workloads
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env["SVDQ_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "ViT-B-32", "--steps", "3",
                        "--warmup", "1", "--no-cpu", "--no-weak"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3 and d["value"] > 0
    shapes = workloads.vit_visual_shapes("ViT-B-32")
    rows = [workloads.numel(shapes[n]) for n in sorted(shapes)]
    parts = shard.partition_lpt(rows, 2)
    ranks = sorted(d["per_rank"]["ranks"], key=lambda e: e["rank"])
    assert [e["rank"] for e in ranks] == [0, 1]
    for e, part in zip(ranks, parts):
        assert e["tensors"] == len(part) and e["sum_rows"] == sum(rows[i] for i in part)
        assert len(e["kernels_ms"]) == 4 and all(x is not None and x > 0 for x in e["kernels_ms"])
    assert sum(e["sum_rows"] for e in ranks) == sum(rows)


_RCCL_SMOKE = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from svdq_amd import shard
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # exactly bench.py's call
assert dist.get_backend() == "nccl"
n = 70001
rg = shard.RaggedGather(n, dev)
assert not rg._staged and rg.send.is_cuda                                   # device buffers, no host staging
side = torch.cuda.Stream()
for step in range(5):                                                       # the pipelined form, as Workload.step uses it
    buf = (torch.arange(n, device=dev) + step).to(torch.uint8)
    rg.run(buf, overlap=True)
    with torch.cuda.stream(side):                                           # "the next step's kernels" on another stream
        torch.ones(1 << 20, device=dev).sum()
rg.finish()
torch.cuda.synchronize()
assert torch.equal(rg.views()[0], ((torch.arange(n, device=dev) + 4).to(torch.uint8)))
G = torch.arange(64, dtype=torch.float64, device=dev).view(8, 8).clone()
want = G.clone()
shard.all_reduce_gram(G)
tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(G, want) and float(tt.item()) == 1.25
dist.destroy_process_group()
print("rccl ok")
'''


@pytest.mark.gpu
def test_rccl_process_group_single_rank_smoke():
    """The RCCL side of the multi-GPU path on the one GPU a test box has: a world of ONE rank over backend "nccl" --
    `init_process_group(device_id=...)`, the planned gather on DEVICE buffers with asynchronous collectives on two
    alternating sets, the fp64 Gram all-reduce, the max-over-ranks reduction and the barrier bench.py times with.  It
    cannot show scaling; it does show that every RCCL call the 8-GPU run makes is accepted by this image's RCCL."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SMOKE, ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.gpu
def test_bench_one_rank_world_over_rccl():
    """bench.py's own `nccl` branch on one GPU (SVDQ_DIST_SINGLE=1: a world of one rank run as a distributed job): RCCL
    process group with device_id, RaggedGather on the device buffer with overlap, barrier / max-over-ranks timing, the
    per-rank report and the separately timed basis gather -- the code the driver's multi-GPU run executes."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("SVDQ_DIST_BACKEND", None)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               SVDQ_DIST_SINGLE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "ViT-B-32", "--steps", "3", "--warmup", "1",
                        "--no-cpu"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert len(d["per_rank"]["ranks"]) == 1 and d["per_rank"]["ranks"][0]["tensors"] == 152
    assert d["basis_gather"] is not None and d["basis_gather"]["ms"] > 0


@pytest.mark.gpu
def test_bench_shard_of_is_rank_zeros_lpt_share():
    """bench.py --shard-of K on one GPU compresses exactly rank 0's LPT share of a K-rank strong-scaling run (the
    rehearsal DESIGN.md section 6 quotes for what one GPU of the 8-GPU run does per step)."""
    import json
    import subprocess
    import sys
    from svdq_amd import workloads, shard
    shapes = workloads.vit_visual_shapes("ViT-B-32")
    rows = [workloads.numel(shapes[n]) for n in sorted(shapes)]
    mine = shard.partition_lpt(rows, 8)[0]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SVDQ_DIST_SINGLE", "SVDQ_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "ViT-B-32", "--shard-of", "8", "--steps", "2",
                        "--warmup", "1", "--no-cpu", "--placement-candidates", "1"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["tensors_rank0"] == len(mine)
    assert f"sum D = {sum(rows[i] for i in mine)} " in d["config"]["workload"]
    assert "share of 8 ranks" in d["config"]["sharding"]
