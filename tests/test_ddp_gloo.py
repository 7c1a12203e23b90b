"""CPU, world_size 2 (gloo): the bucketed gradient reducer averages gradients across ranks bucket by bucket, honours
no_sync() gradient accumulation, and every rank ends with identical values."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rga3.parallel.ddp import GradBucketReducer

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 3))
    red = GradBucketReducer(net.parameters(), bucket_mb=0.0003)  # ~300 B buckets -> several buckets
    assert len(red.buckets) >= 3
    xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(10 * r + m)) for r in range(world) for m in range(2)]
    my = xs[2 * rank: 2 * rank + 2]
    red.begin_step()
    with red.no_sync():
        red.begin_micro_step()
        net(my[0]).pow(2).sum().backward()
    red.begin_micro_step()
    net(my[1]).pow(2).sum().backward()
    red.finish()
    got = [red.grad_view(p).clone() for p in net.parameters()]
    red.remove()
    # reference: average over ranks of the per-rank accumulated gradients
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(world):
        net.zero_grad()
        for m in range(2):
            net(xs[2 * r + m]).pow(2).sum().backward()
        for a, p in zip(ref, net.parameters()):
            a += p.grad / world
    ok = all(torch.allclose(a, b, atol=1e-5) for a, b in zip(got, ref))
    # a parameter the loss never reaches (its bucket never counts down during backward) + accumulation: the bucket is still exchanged in
    # finish(), the used parameters accumulate over both micro-steps, the unused one reads as zeros -- also on a second optimizer step
    torch.manual_seed(1)
    used, unused = torch.nn.Linear(8, 4), torch.nn.Linear(8, 4)
    ref2 = [torch.zeros_like(p) for p in used.parameters()]          # reference first: no reducer hooks attached yet
    for r in range(world):
        used.zero_grad()
        for m in range(2):
            used(xs[2 * r + m]).pow(2).sum().backward()
        for a, p in zip(ref2, used.parameters()):
            a += p.grad / world
    used.zero_grad()
    red2 = GradBucketReducer(list(used.parameters()) + list(unused.parameters()), bucket_mb=1.0)   # ONE bucket holding all four tensors
    for step in range(2):
        red2.begin_step()
        with red2.no_sync():
            red2.begin_micro_step()
            used(my[0]).pow(2).sum().backward()
        red2.begin_micro_step()
        used(my[1]).pow(2).sum().backward()
        red2.finish()
        ok = ok and all(torch.allclose(red2.grad_view(p), a, atol=1e-5) for p, a in zip(used.parameters(), ref2))
        ok = ok and all(float(red2.grad_view(p).abs().max()) == 0.0 for p in unused.parameters())
    red2.remove()
    q.put((rank, ok, [g.sum().item() for g in got]))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]  # identical on both ranks
